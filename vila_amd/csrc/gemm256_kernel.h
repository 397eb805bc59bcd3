// 256x256x64 bf16 MFMA GEMM kernel template (see gemm256.hip for the design notes); shared by gemm256.hip (forward layouts) and
// gemm256_cm.hip (contraction-major operands for dgrad / wgrad, schedule variants).
//
// Operand storage:
//   CC ("contraction-contiguous", the forward layout): X[rows][K], K contiguous  — nn.Linear weights [out,in] and activations [tok,ch]
//   CM ("contraction-major"):                          X[K][rows], rows contiguous — what dgrad / wgrad meet when they consume the SAME
//       tensors in place: dX = dY . W reads W[N,K] with the contraction over its ROW index; dW = dY^T . X reads dY[T,N] and X[T,K]
//       with the contraction over the token index.  No transposed copies are made: the LDS image of a CM half-tile is [64 k][128 rows]
//       and the MFMA fragments (8 consecutive k per lane) come out of `ds_read_b64_tr_b16` (two reads per fragment), the same hardware
//       transpose the attention kernels use for V.
//   LDS images are lane-linear for the LDS-DMA; bank conflicts are removed by permuting the per-lane SOURCE address and applying the
//   same involution on the reads (guide rule 21): CC: 16-B slot ^= (row>>1)&7;  CM: 32-B pair ^= (k&3) | ((k>>3)&1)<<2, which puts the
//   eight k-rows a 32-lane tr-read cycle touches on eight different 32-B bank groups.
//
// SCHED (how the 8 LDS-DMA instructions per thread and K-tile are issued):
//   0  one tile ahead: top of tile t {vmcnt(0); barrier; issue all of tile t+1}                                  (round-1 kernel)
//   1  two tiles ahead, refill-after-use: top of tile t {vmcnt(8); barrier}; a barrier in front of phases 1,2,3; the pieces of tile
//      t+2 are issued into the buffer tile t is being read from as soon as every wave has finished with them
//      (phase 1: A rows 0-63 of both halves; phase 2: both B halves; phase 3: A rows 64-127).  The wait never drains the queue.
//   2  as 1 but one extra barrier only (before phase 3) and all 8 pieces of tile t+2 issued there
//   3  register-pipelined fragments, one barrier per tile (forward default)
//   5  role-split 8-barrier schedule: the two wave groups run one barrier interval apart (see the K loop)
//   6  as 5 with the fragment reads retired in front of the barrier and two DMA pieces per phase
//   7  role split with TWO phases of 32 MFMAs per K-tile (4 barriers): the default
//   9  ablation: schedule 0 without any DMA inside the loop (wrong results; bounds what hiding the loads completely would buy)
#pragma once
#include <stdlib.h>
#include "kernels.h"

#define T256_BK 64
#define T256_STG 68

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
typedef __attribute__((address_space(3))) bf16x4v lds_bf16x4;

constexpr int T256_CC_SCHED = 7;   // default schedule of every layout: role split, two phases of 32 MFMAs per K-tile
static __device__ __attribute__((aligned(16))) unsigned int g_zero_chunk[4];   // K-tail source (zero-initialised); one per translation unit (no RDC)

// MODE: 0 = bf16 out (bias/GELU/residual), 1 = fp32 out, 2 = gate/up fused (bf16 out), 3 = split-K fp32 slab (raw accumulators),
//       4 = gate/up split-K: raw gate and up accumulators into two fp32 planes per K-slice (the tail round of an under-filled grid)
//       5 = split-K into COMPACT per-tile slabs [K-slice][tile of this launch][256][256] fp32 (the tail tiles behind whole rounds)
// tile0 = first tile id of this launch (a GEMM may be issued as "full rounds" + "split tail"), col0 = first output column of the slab
// BM = output rows per tile: 256, or 192 (role-split schedules, A in the forward layout): M = 3076 x N = 3584 is 13 x 14 = 182 tiles of
// 256 rows — one round on 71 % of the CUs — but 17 x 14 = 238 tiles of 192 rows, each 3/4 of the MFMAs.  The LDS image keeps its 128-row
// half-tile slots (rows 96..127 of a half are loaded and not used), each wave owns 96 rows = 6 A fragments.  Measured: the K-loop's pace
// is set by its barrier intervals more than by its MFMA count, so 3/4 of the MFMAs buy 2-9 % (o_proj 92 -> 84 us, down_proj 424 -> 417 us),
// not 25 %; used for the forward layout only.
// EX ("ninth A fragment", round 3): M = 256 k + r with 1 <= r <= 16 is tiled as k row tiles, and the LAST row tile carries the r leftover
// rows as a 17th 16-row fragment: its blocks stage 16 more A rows per K-tile (2 KB, one extra LDS-DMA instruction on waves 0 and 1) and every
// wave spends 4 more MFMAs per K-tile on two of its four B fragments (wave row wr takes fragments wr and wr + 2, so the fused gate/up
// pairing stays inside a wave).  The prompt of the benchmark is 3 x 256 + 1 tokens: without this, one row costs either a fourth row-tile
// round or a second pass over the MLP weights through the decode GEMVs (round 1/2: 1.9 ms of an 18.4 ms TTFT).
// ---- tile order (round 3) ---------------------------------------------------------------------------------------------------------------
// A tile id is decoded either tm-fastest (id = tn * tiles_m + tm: the order of rounds 1-2) or GROUPED: columns in groups of `grp`, inside a
// group tn fastest, then tm — so that the 32 tiles an XCD works on at one time (consecutive ids after xcd_remap) form an 8 x 4 patch
// instead of a 32 x 1 strip.  Why: each XCD has its own 4-MB L2.  In a strip the 32 tiles share ONE B tile and stream 32 different A
// tiles, and with tiles_m >> 32 nothing of A is still there when the next column comes by: the PMC counters of the SFT wgrad shapes
// (profiles/r03_pmc_gemm_sft.txt) show A fetched from the fabric once per TILE — gate wgrad 1.74 GB of L2 fills for 138 MB of operands.
// A patch shares every A tile 4 ways and every B tile 8 ways: (8 + 4) operand tiles per 32 instead of 33.  Both orders are bijections on
// [0, tiles), so launches over an id range (whole rounds + K-sliced tail) and the tail's reduce kernel work with either.
__host__ __device__ inline void gemm256_tile_of(int id, int tiles_m, int tiles_n, int grp, int& tm, int& tn) {
    if (grp > 1) {
        const int per = grp * tiles_m, g0 = (id / per) * grp, w = id % per;
        const int gs = tiles_n - g0 < grp ? tiles_n - g0 : grp;
        tn = g0 + w % gs; tm = w / gs;
    } else {
        tm = id % tiles_m; tn = id / tiles_m;
    }
}
extern int g_gemm256_group;        // tuning hook (vila_gemm_force_group): -1 = the rule below, 0 = tm-fastest everywhere, n = groups of n
// grouped order where a strip would be long (tiles_m > 16); `gateup` = a fused gate/up launch over part of the grid (tail policy): never
static inline int gemm256_group(int tiles_m, int tiles_n, bool gateup) {
    if (gateup || tiles_n < 2) return 0;
    if (g_gemm256_group >= 0) return g_gemm256_group;
    return tiles_m > 16 ? 4 : 0;
}

// (Measured and removed in round 5: an "epilogue prefetch" variant that requested a store pass's residual words ahead of the pass — the ISA shows
// `global_load_dwordx2 -> s_waitcnt vmcnt(0) -> add -> global_store` 32 times per wave and tile with a residual.  Bit-identical results, and no
// effect on the SFT step: 194.9 / 196.1 ms without, 195.3 / 216.0 with; profiles/r05_second_call_ab.log.)
template <int MODE, int EPI, bool ACM, bool BCM, int SCHED, int BM = 256, bool EX = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p, int tiles_m, int k_tiles_per_split, int tile0, int col0, int grp) {
    static_assert(BM == 256 || (BM == 192 && !ACM && (SCHED == 5 || SCHED == 6 || SCHED == 7) && MODE == 0), "192-row tiles: role-split schedules, forward-layout A");
    static_assert(!EX || (!ACM && SCHED == 7 && BM == 256 && MODE != 5), "the extra row fragment: forward-layout A, default schedule, 256-row tiles");
    constexpr int NA = BM / 64;                        // A fragments per quadrant (4, or 3 with 192-row tiles)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HALF_BYTES = 128 * 64 * 2;           // 16 KB
    constexpr int BUF_BYTES = 4 * HALF_BYTES;          // A_lo, A_hi, B_lo, B_hi
    constexpr bool GU = (MODE == 2 || MODE == 4);
    constexpr bool SPLIT = (MODE == 3 || MODE == 4 || MODE == 5);
    constexpr int BN_OUT = GU ? 128 : 256;
    static_assert(!(GU && (ACM || BCM)), "gate/up fusion only for the forward layouts");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, gridDim.x) + tile0;
    const int M = p.M, N = p.N;
    int tm, tn;
    gemm256_tile_of(id, tiles_m, (N + BN_OUT - 1) / BN_OUT, grp, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN_OUT;
    const bool has9 = EX && (tm == tiles_m - 1) && (M > tiles_m * 256);        // block-uniform: this row tile carries rows m0 + 256 .. M - 1
    constexpr int A9_BASE = 2 * 4 * 128 * 64 * 2;                               // behind the two K-tile buffers: 2 x [16 rows][64 k] bf16

    // ---- DMA source offsets ----
    // CC: thread's chunk c = tid + 512*i of a half-tile: row = c >> 3 (+64 i), LDS slot = c & 7 holds global k-chunk (c&7) ^ ((row>>1)&7)
    // CM: chunk c = tid + 512*i: k = c >> 4 (+32 i), LDS slot s = c & 15 holds the 8 rows of global chunk (((s>>1) ^ g(k)) << 1) | (s&1)
    uint32_t aoff[2][2], boff[2][2];                   // [half][round] element offsets (CC: row start + k chunk; CM: k row + row chunk)
    bool b_up[2][2];
    const int srow = tid >> 3;
    const int kch = (tid & 7) ^ ((srow >> 1) & 7);
    const int cm_k = tid >> 4;                         // + 32 i
    const int cm_g = (cm_k & 3) | (((cm_k >> 3) & 1) << 2);       // invariant under k += 32
    const int cm_ch = ((((tid & 15) >> 1) ^ cm_g) << 1) | (tid & 1);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (ACM) {
                int gm = m0 + h * 128 + cm_ch * 8; gm = gm + 8 <= M ? gm : (M >= 8 ? M - 8 : 0);      // rows beyond M: any valid rows (masked at the store)
                aoff[h][i] = (uint32_t)(cm_k + 32 * i) * (uint32_t)p.lda + gm;
            } else {
                int gm = m0 + h * (BM / 2) + srow + 64 * i; gm = gm < M ? gm : M - 1;
                aoff[h][i] = (uint32_t)gm * (uint32_t)p.lda + kch * 8;
            }
            b_up[h][i] = false;
            if constexpr (BCM) {
                int gn = n0 + h * 128 + cm_ch * 8; gn = gn + 8 <= N ? gn : (N >= 8 ? N - 8 : 0);
                boff[h][i] = (uint32_t)(cm_k + 32 * i) * (uint32_t)p.ldw + gn;
            } else {
                const int rb = h * 128 + srow + 64 * i;    // row of the 256-row B tile; wave column = rb >> 6
                int gn;
                if (GU) { gn = n0 + (rb >> 6) * 32 + (rb & 31); b_up[h][i] = (rb & 32) != 0; }
                else { gn = n0 + rb; }
                gn = gn < N ? gn : N - 1;
                boff[h][i] = (uint32_t)gn * (uint32_t)p.ldw + kch * 8;
            }
        }
    const int lds_lane_base = __builtin_amdgcn_readfirstlane(wave * 1024);   // wave-uniform: 64 lanes x 16 B per DMA
    const int kt0 = SPLIT ? blockIdx.y * k_tiles_per_split : 0;

    // K-tail source, pinned into an SGPR pair ONCE: left to itself the compiler re-loads the symbol's address from the GOT in front of every
    // DMA (s_load + s_waitcnt lgkmcnt(0)), and that wait also drains every LDS fragment read in flight
    const bf16_t* zero_src = (const bf16_t*)g_zero_chunk;
    asm volatile("" : "+s"(zero_src));
    // One piece = one LDS-DMA per thread (8 KB): which in {A, B}, half h, round i.
    // Round 5: the address is formed as  WAVE-UNIFORM 64-bit base (operand pointer + this K-tile's offset, SALU)  +  per-lane 32-bit BYTE offset
    // (fixed for the whole kernel), which the compiler encodes as the `saddr + voffset` form of global_load_lds: m0 + the load, 2 instructions per
    // piece.  Before, the ISA of the K loop carried 7 per piece — v_lshl_add_u64 for the 64-bit per-lane pointer and FOUR v_cndmask for the K-tail
    // select against the zero chunk, which the compiler had if-converted into every tile although only a ragged LAST tile needs it (K % 64 != 0:
    // the wgrads' T = 3076, the tower's K = 4304) — 8 pieces x 5 VALU per wave and K-tile inside the read / issue phases that pace the loop.
    // A ragged last tile now takes a real branch (the asm statement behind its load keeps the two loads from being merged back into a select).
    // gemm256_supported() bounds every element offset by 2^31, so the byte offsets fit 32 bits.
    uint32_t aoffb[2][2], boffb[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) { aoffb[h][i] = aoff[h][i] * 2u; boffb[h][i] = boff[h][i] * 2u; }
    const bool up_w = GU && (__builtin_amdgcn_readfirstlane(wave) & 4) != 0;     // == b_up[h][i] for every piece of this wave (bit 5 of the B-tile row)
    // pieces (h, i) for h in [h0, h1), i in [i0, i1) of tile t: ONE block-uniform branch for the group
    auto issue_pieces = [&](int t, int buf, bool is_b, int h0, int h1, int i0, int i1) {
        const int k0 = (kt0 + t) * T256_BK;
        const bool cm = is_b ? BCM : ACM;
        const bf16_t* base = is_b ? (up_w ? p.W2 : p.W) : p.A;
        const int64_t ld = is_b ? p.ldw : p.lda;
        const char* sb = (const char*)(cm ? base + (int64_t)k0 * ld : base + k0);          // wave-uniform
        char* dst0 = smem + buf * BUF_BYTES + lds_lane_base + (is_b ? 2 : 0) * HALF_BYTES;
        if (k0 + T256_BK > p.K) {                                                           // ragged last K-tile (block-uniform)
#pragma unroll
            for (int h = h0; h < h1; ++h)
#pragma unroll
                for (int i = i0; i < i1; ++i) {
                    const bool kin = cm ? (k0 + cm_k + 32 * i < p.K) : (k0 + kch * 8 < p.K);
                    const uint32_t vo = is_b ? boffb[h][i] : aoffb[h][i];
                    const char* src = kin ? sb + vo : (const char*)zero_src;
                    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(dst0 + h * HALF_BYTES + i * 8192), 16, 0, 0);
                }
            asm volatile("" ::: "memory");                                                  // (keeps this branch a branch: see above)
        } else {
#pragma unroll
            for (int h = h0; h < h1; ++h)
#pragma unroll
                for (int i = i0; i < i1; ++i) {
                    uint32_t& vo = is_b ? boffb[h][i] : aoffb[h][i];                        // per lane, bytes
                    asm volatile("" : "+v"(vo));                // in place, no instruction: the zero-extension must stay in THIS block (hoisted out
                    __builtin_amdgcn_global_load_lds((gbl_void*)(sb + vo), (lds_void*)(dst0 + h * HALF_BYTES + i * 8192), 16, 0, 0);   // of the loop as a 64-bit pair it hides the saddr form)
                }
        }
    };
    auto issue_piece = [&](int t, int buf, bool is_b, int h, int i) { issue_pieces(t, buf, is_b, h, h + 1, i, i + 1); };
    // the extra fragment's 16 rows x 64 k = 128 chunks: one DMA instruction on wave 0 and one on wave 1, same slot swizzle as a half-tile
    uint32_t a9off = 0; int a9kch = 0;
    if constexpr (EX) {
        const int c9 = (wave & 1) * 64 + lane, row9 = c9 >> 3;
        a9kch = (c9 & 7) ^ ((row9 >> 1) & 7);
        int gm9 = m0 + 256 + row9; gm9 = gm9 < M ? gm9 : M - 1;
        a9off = (uint32_t)gm9 * (uint32_t)p.lda + a9kch * 8;
    }
    uint32_t a9offb = a9off * 2u;
    const int a9_lane_base = __builtin_amdgcn_readfirstlane((wave & 1) * 1024);
    auto issue_a9 = [&](int t, int buf) {
        if constexpr (EX) {
            if (has9 && wave < 2) {
                const int k0 = (kt0 + t) * T256_BK;
                char* dst = smem + A9_BASE + buf * 2048 + a9_lane_base;
                if (k0 + T256_BK > p.K) {                  // ragged last K-tile: per-lane select (a branch, like issue_pieces)
                    const bf16_t* src = (k0 + a9kch * 8 < p.K) ? p.A + a9off + k0 : zero_src;
                    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)dst, 16, 0, 0);
                    asm volatile("" ::: "memory");
                } else {                                   // saddr + 32-bit voffset form
                    asm volatile("" : "+v"(a9offb));
                    __builtin_amdgcn_global_load_lds((gbl_void*)((const char*)(p.A + k0) + a9offb), (lds_void*)dst, 16, 0, 0);
                }
            }
        }
    };
    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) { issue_piece(t, buf, false, h, i); issue_piece(t, buf, true, h, i); }      // (prologue only: order A, B per piece as the counted waits expect)
        issue_a9(t, buf);
    };

    // ---- fragment read offsets (bytes inside a half-tile) ----
    const int swr = (l15 >> 1) & 7;
    int foff[2];                                       // CC: row l15 of a 16-row fragment, k-step ks
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = l15 * 128 + (((ks * 4 + lg) ^ swr) << 4);
    // CM: k row = ks*32 + lg*8 + (l15>>2) (+4 for the second read), 32-B pair P ^ g with g = (l15>>2) | (lg&1)<<2, 8-B column (l15&3)
    const int cm_rg = (l15 >> 2) | ((lg & 1) << 2);
    const int cm_base = (lg * 8 + (l15 >> 2)) * 256 + (l15 & 3) * 8;
    const int a_half = wr, b_half = wc >> 1, b_row0 = (wc & 1) * 64;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 acc9[2];                                     // EX: the extra fragment x this wave's B fragments (wr, wr + 2)
    acc9[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc9[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment f (16 rows at row16 = f) of a half-tile at `base`, k-step ks
    auto frag_cc = [&](const char* base, int row0, int ks) -> bf16x8 { return *(const bf16x8*)(base + row0 * 128 + foff[ks]); };
    // SCHED 7 reads its contraction-major fragments with the ASM form of the transpose read (common.h): the builtin is treated as a
    // possible LDS write and gets an `s_waitcnt vmcnt(0)` in front of it whenever LDS-DMA pieces are in flight — i.e. the queue this
    // schedule keeps 4-8 pieces deep was drained in every phase of the dgrad / wgrad kernels (round 2: MfmaUtil 0.39-0.40 against
    // 0.54 for the forward layout, whose ds_read_b128 are not affected).  The asm results are retired by `retire()` below.
    constexpr bool ASM_TR = (SCHED == 7) && (ACM || BCM);
    auto frag_cm = [&](const char* base, int row0, int ks) -> bf16x8 {
        const char* q = base + cm_base + ks * 32 * 256 + ((((row0 >> 4) ^ cm_rg) & 7) << 5);
        if constexpr (ASM_TR) {
            const uint32_t a = lds_addr(q);
            const u32x2 lo = ds_read_tr16_b64<0>(a), hi = ds_read_tr16_b64<4 * 256>(a);
            u32x4 w; w[0] = lo[0]; w[1] = lo[1]; w[2] = hi[0]; w[3] = hi[1];
            return __builtin_bit_cast(bf16x8, w);
        } else {
            const bf16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)q);
            const bf16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(q + 4 * 256));
            return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    auto fragA = [&](const char* base, int row0, int ks) -> bf16x8 { if constexpr (ACM) return frag_cm(base, row0, ks); else return frag_cc(base, row0, ks); };
    auto fragB = [&](const char* base, int row0, int ks) -> bf16x8 { if constexpr (BCM) return frag_cm(base, row0, ks); else return frag_cc(base, row0, ks); };
    // SCHED 7: the k-step's 8 KB stride travels in the instruction's offset field (an asm read cannot have it folded by the compiler)
    auto frag_cm7 = [&](const char* base, int row0, auto ksc) -> bf16x8 {
        constexpr int KS = decltype(ksc)::value;
        const uint32_t a = lds_addr(base + cm_base + ((((row0 >> 4) ^ cm_rg) & 7) << 5));
        const u32x2 lo = ds_read_tr16_b64<KS * 32 * 256>(a), hi = ds_read_tr16_b64<KS * 32 * 256 + 4 * 256>(a);
        u32x4 w; w[0] = lo[0]; w[1] = lo[1]; w[2] = hi[0]; w[3] = hi[1];
        return __builtin_bit_cast(bf16x8, w);
    };
    auto fragA7 = [&](const char* base, int row0, auto ksc) -> bf16x8 {
        if constexpr (ACM) return frag_cm7(base, row0, ksc); else return frag_cc(base, row0, decltype(ksc)::value);
    };
    auto fragB7 = [&](const char* base, int row0, auto ksc) -> bf16x8 {
        if constexpr (BCM) return frag_cm7(base, row0, ksc); else return frag_cc(base, row0, decltype(ksc)::value);
    };

    const int kt_all = (p.K + T256_BK - 1) / T256_BK;
    const int nt = SPLIT ? ((kt_all - kt0) < k_tiles_per_split ? (kt_all - kt0) : k_tiles_per_split) : kt_all;   // last slice may be shorter
    if constexpr (SCHED == 3) {
        // Register-pipelined fragments: every ds_read is issued one phase (16 MFMAs) before its first use, ONE barrier per K-tile.
        //   quadrant order (A0,B0) (A0,B1) (A1,B0) (A1,B1); A1 / B1 of tile t are requested at the top of tile t, A0 / B0 of tile t+1 during
        //   the last quadrant of tile t — after the tile's only barrier, which is also where tile t+1's DMA is waited for.
        //   The 8 DMA pieces of tile t+2 are issued in one burst right after that barrier (threading them through the MFMA stream one
        //   piece per 4 MFMAs via inline asm was measured too: no faster on any SFT shape, 8 % slower on K = 18944 — dropped).
        static_assert(!(SCHED == 3 && (ACM || BCM)), "SCHED 3 is implemented for the forward layout");
        bf16x8 A0[4][2], A1[4][2], B0[2][2], B1[2][2];
        issue_tile(0, 0);
        if (nt > 1) { issue_tile(1, 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const char* cA = smem + a_half * HALF_BYTES;
            const char* cB = smem + (2 + b_half) * HALF_BYTES;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) B0[j][ks] = frag_cc(cB, b_row0 + j * 16, ks);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) A0[i][ks] = frag_cc(cA, i * 16, ks);
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        // 16 MFMAs of one quadrant
        auto quadrant = [&](bf16x8 (&A)[4][2], bf16x8 (&Bf)[2][2], int ai, int bj) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ai + i][bj + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i][ks], Bf[j][ks], acc[ai + i][bj + j], 0, 0, 0);
        };
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            const char* cA = smem + buf * BUF_BYTES + a_half * HALF_BYTES;
            const char* cB = smem + buf * BUF_BYTES + (2 + b_half) * HALF_BYTES;
            // request B1, A1 of this tile, then quadrant (0,0) on the fragments that are already in registers
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) B1[j][ks] = frag_cc(cB, b_row0 + 32 + j * 16, ks);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) A1[i][ks] = frag_cc(cA, 64 + i * 16, ks);
            quadrant(A0, B0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);              // keep the quadrants in order: the first one must not wait for the new reads
            quadrant(A0, B1, 0, 2);
            __builtin_amdgcn_sched_barrier(0);
            quadrant(A1, B0, 4, 0);
            if (t + 1 < nt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of tile t+1 have landed
                __builtin_amdgcn_s_barrier();                        // everyone's have; every wave has its fragments of tile t
                asm volatile("" ::: "memory");
                if (t + 2 < nt) issue_tile(t + 2, buf);
                const char* nA = smem + (buf ^ 1) * BUF_BYTES + a_half * HALF_BYTES;
                const char* nB = smem + (buf ^ 1) * BUF_BYTES + (2 + b_half) * HALF_BYTES;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) B0[j][ks] = frag_cc(nB, b_row0 + j * 16, ks);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) A0[i][ks] = frag_cc(nA, i * 16, ks);
            }
            quadrant(A1, B1, 4, 2);
            // the A0 / B0 reads of the next tile were issued 16 MFMAs ago: retire them HERE (free), so that the compiler's wait-count
            // bookkeeping carries no pending LDS read over the back edge (it would otherwise wait for ALL reads before the first MFMA)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (SCHED == 7) {
        // Role split with TWO phases of 32 MFMAs per K-tile (4 barriers per tile instead of 8).  Measured premise: a 192-row tile (12
        // MFMAs per phase instead of 16) runs the SCHED 6 loop in the same time per K-tile — the barrier intervals, not the MFMA count,
        // set its pace.  Phase 0: reads B0, B1, A0 (16 fragments reads), quadrants (A0,B0) (A0,B1); phase 1: reads A1, quadrants (A1,B1)
        // (A1,B0).  Fragment reads are retired in front of the section's barrier (as in SCHED 6).  (ONE phase of 64 MFMAs — 2 barriers per
        // tile, all 24 fragments live, asymmetric DMA roles — was built and measured too: 96 fragment registers push the kernel to 256 VGPRs
        // with spills and it runs 15-27 % slower than this schedule.)  Barrier arithmetic, tile t, phase p:
        // group 0 reads / issues in interval 4t+2p+1 and multiplies in 4t+2p+2, group 1 one later.
        //   WAR: all B reads of tile t (phase 0) are retired by every wave before barrier 4t+2, the A reads (phase 1) before 4t+4
        //        -> B halves of tile t+2 are staged in phase 1 of tile t (group 0: after 4t+2), A halves in phase 0 of tile t+1 (after 4t+4)
        //   RAW: the counted wait sits in phase 1 in front of its barrier (4t+3 / 4t+4) and leaves the 4 pieces issued in the same
        //        section (B of tile t+2) in flight; the first reads of tile t+1 follow barrier 4t+4 (group 0) / 4t+5 (group 1).
        issue_tile(0, 0);
        if (nt > 1) { issue_tile(1, 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wr == 1) __builtin_amdgcn_s_barrier();
        bf16x8 af[4][2], bf0[2][2], bf1[2][2];
        // lgkmcnt(0) with the fragment registers tied to it: results of the asm transpose reads must not be consumed (nor the MFMAs
        // that consume them be scheduled) in front of this wait; for compiler-tracked reads the tie is a no-op
        auto retire_a = [&]() {
            if constexpr (ASM_TR) {
                if constexpr (NA == 4)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), "+v"(af[2][1]),
                                 "+v"(af[3][0]), "+v"(af[3][1]) :: "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), "+v"(af[2][1])
                                 :: "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        };
        auto retire_b = [&]() {
            if constexpr (ASM_TR)
                asm volatile("" : "+v"(bf0[0][0]), "+v"(bf0[0][1]), "+v"(bf0[1][0]), "+v"(bf0[1][1]), "+v"(bf1[0][0]), "+v"(bf1[0][1]),
                             "+v"(bf1[1][0]), "+v"(bf1[1][1]));
        };
        auto enter_mfma = [&](bool with_b) {
            __builtin_amdgcn_sched_barrier(0);
            retire_a();                                      // s_waitcnt lgkmcnt(0): every outstanding LDS read of this wave
            if (with_b) retire_b();                          // (same wait: only ties the B fragments behind it)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
        };
        auto leave_mfma = [&]() {
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mfma16 = [&](bf16x8 (&A)[4][2], bf16x8 (&Bf)[2][2], int ai, int bj) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ai + i][bj + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i][ks], Bf[j][ks], acc[ai + i][bj + j], 0, 0, 0);
        };
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            const char* cA = smem + buf * BUF_BYTES + a_half * HALF_BYTES;
            const char* cB = smem + buf * BUF_BYTES + (2 + b_half) * HALF_BYTES;
            // phase 0: all B fragments, A rows 0..63; the A halves of tile t+1 go into the other buffer (tile 1's came with the prologue)
            static_for<0, 2>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
#pragma unroll
                for (int j = 0; j < 2; ++j) { bf0[j][ks] = fragB7(cB, b_row0 + j * 16, ksc); bf1[j][ks] = fragB7(cB, b_row0 + 32 + j * 16, ksc); }
            });
            static_for<0, 2>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
#pragma unroll
                for (int i = 0; i < NA; ++i) af[i][ks] = fragA7(cA, i * 16, ksc);
            });
            bf16x8 a9[2];
            if constexpr (EX) {
                if (has9) {
                    const char* c9 = smem + A9_BASE + buf * 2048;
                    a9[0] = frag_cc(c9, 0, 0); a9[1] = frag_cc(c9, 0, 1);
                }
            }
            if (t >= 1 && t + 1 < nt) {
                issue_pieces(t + 1, buf ^ 1, false, 0, 2, 0, 2);
                issue_a9(t + 1, buf ^ 1);                  // with the A halves: older than the B halves the counted wait leaves in flight
            }
            enter_mfma(true); mfma16(af, bf0, 0, 0); mfma16(af, bf1, 0, 2);
            if constexpr (EX) {
                if (has9) {                                  // 4 MFMAs: the extra 16 rows x this wave's B fragments wr and wr + 2
                    if (wr == 0) {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            acc9[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a9[ks], bf0[0][ks], acc9[0], 0, 0, 0);
                            acc9[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a9[ks], bf1[0][ks], acc9[1], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            acc9[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a9[ks], bf0[1][ks], acc9[0], 0, 0, 0);
                            acc9[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a9[ks], bf1[1][ks], acc9[1], 0, 0, 0);
                        }
                    }
                }
            }
            leave_mfma();
            // phase 1: A rows 64..127; the B halves of tile t+2 into THIS buffer, then the counted wait for tile t+1
            static_for<0, 2>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
#pragma unroll
                for (int i = 0; i < NA; ++i) af[i][ks] = fragA7(cA, NA * 16 + i * 16, ksc);
            });
            if (t + 2 < nt) {
                issue_pieces(t + 2, buf, true, 0, 2, 0, 2);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else if (t + 1 < nt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            enter_mfma(false); mfma16(af, bf1, NA, 2); mfma16(af, bf0, NA, 0); leave_mfma();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else if constexpr (SCHED == 5 || SCHED == 6) {
        // Role-split schedule (the guide's 8-phase structure): the K-tile is four phases {fragment reads + LDS-DMA issue | barrier |
        // 16 MFMAs | barrier}, and the two wave groups (wr = 0 / 1: ONE wave of each on every SIMD) run ONE barrier interval apart, so
        // that while a SIMD's first wave is in its MFMA section (s_setprio 1) the second is issuing ds_reads / global_load_lds and vice
        // versa.  In the lock-step schedules both waves of a SIMD issue their 8 LDS-DMA instructions (~64 issue cycles each) at the same
        // moment and the matrix pipe idles for that long every K-tile (SCHED 9 ablation: +25-40 % with the DMA removed).
        //
        // Barrier arithmetic (group 1 executes one extra barrier before the loop, group 0 one after it: equal totals).  With barriers
        // numbered globally, tile t, phase p:  group 0 reads/issues in interval 8t+2p+1 and multiplies in 8t+2p+2, group 1 one later.
        //   WAR: the last B-fragment reads of tile t (phase 1) are retired by every wave before barrier 8t+5, the last A reads (phase 2)
        //        before 8t+7  ->  the B halves of tile t's buffer are re-staged (tile t+2) in phase 3 of tile t (group 0: after 8t+6),
        //        the A halves in phase 0 of tile t+1 (group 0: after 8t+8).
        //   RAW: everything of tile t+1 is retired by the counted wait in phase 3 of tile t, in front of that phase's first barrier
        //        (8t+7 / 8t+8); tile t+1's first reads follow barrier 8t+8 (group 0) / 8t+9 (group 1).  The wait leaves the 4 pieces
        //        issued in the same phase (B halves of tile t+2) in flight: the queue is never drained inside the loop.
        issue_tile(0, 0);
        if (nt > 1) { issue_tile(1, 1); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wr == 1) __builtin_amdgcn_s_barrier();
        bf16x8 af[4][2], bf0[2][2], bf1[2][2];
        // SCHED 6 = SCHED 5 with (a) the fragment reads retired BEFORE the section's barrier (lgkmcnt(0) in front of it): the proof that
        // every wave is done with a half-tile then comes one barrier earlier (B halves free after 8t+4, A halves after 8t+6), which lets
        // (b) the 8 LDS-DMA pieces go out two per phase: B_lo / B_hi of tile t+2 in phases 2 / 3 of tile t, A_lo / A_hi in phases 0 / 1
        // of tile t+1; the counted wait in phase 3 still leaves the 4 newest pieces in flight.
        constexpr bool EARLY = (SCHED == 6);
        auto enter_mfma = [&]() {                       // end of a read/issue section
            __builtin_amdgcn_sched_barrier(0);
            if (EARLY) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (EARLY) asm volatile("" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
        };
        auto leave_mfma = [&]() {
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mfma16 = [&](bf16x8 (&A)[4][2], bf16x8 (&Bf)[2][2], int ai, int bj) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < NA; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[ai + i][bj + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i][ks], Bf[j][ks], acc[ai + i][bj + j], 0, 0, 0);
        };
        for (int t = 0; t < nt; ++t) {
            const int buf = t & 1;
            const char* cA = smem + buf * BUF_BYTES + a_half * HALF_BYTES;
            const char* cB = smem + buf * BUF_BYTES + (2 + b_half) * HALF_BYTES;
            // phase 0: B cols 0..31, A rows 0..63; the A halves of tile t+1 go into the other buffer (tile 1's came with the prologue)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bf0[j][ks] = fragB(cB, b_row0 + j * 16, ks);
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[i][ks] = fragA(cA, i * 16, ks);
            if (t >= 1 && t + 1 < nt) {
#pragma unroll
                for (int h = 0; h < (EARLY ? 1 : 2); ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i) issue_piece(t + 1, buf ^ 1, false, h, i);
            }
            enter_mfma(); mfma16(af, bf0, 0, 0); leave_mfma();
            // phase 1: B cols 32..63
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) bf1[j][ks] = fragB(cB, b_row0 + 32 + j * 16, ks);
            if (EARLY && t >= 1 && t + 1 < nt) { issue_piece(t + 1, buf ^ 1, false, 1, 0); issue_piece(t + 1, buf ^ 1, false, 1, 1); }
            enter_mfma(); mfma16(af, bf1, 0, 2); leave_mfma();
            // phase 2: A rows 64..127
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) af[i][ks] = fragA(cA, NA * 16 + i * 16, ks);
            if (EARLY && t + 2 < nt) { issue_piece(t + 2, buf, true, 0, 0); issue_piece(t + 2, buf, true, 0, 1); }
            enter_mfma(); mfma16(af, bf1, NA, 2); leave_mfma();
            // phase 3: no reads; the B halves of tile t+2 into THIS buffer, then the counted wait for tile t+1
            if (t + 2 < nt) {
#pragma unroll
                for (int h = (EARLY ? 1 : 0); h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i) issue_piece(t + 2, buf, true, h, i);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else if (t + 1 < nt) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            enter_mfma(); mfma16(af, bf0, NA, 0); leave_mfma();
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
    constexpr bool DEEP = (SCHED == 1 || SCHED == 2);
    issue_tile(0, 0);
    if (DEEP && nt > 1) issue_tile(1, 1);
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (DEEP) {
            if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // tile t landed; tile t+1's 8 pieces may still fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // own DMA of tile t has landed
        }
        __builtin_amdgcn_s_barrier();                               // everyone's has; everyone left tile t-1
        asm volatile("" ::: "memory");                              // keep the LDS reads of tile t below the barrier
        if (SCHED == 0 && t + 1 < nt) issue_tile(t + 1, buf ^ 1);
        const bool more = DEEP && (t + 2 < nt);
        const char* cA = smem + buf * BUF_BYTES + a_half * HALF_BYTES;
        const char* cB = smem + buf * BUF_BYTES + (2 + b_half) * HALF_BYTES;
        // CC half-tiles are [128 rows][64 k]: the wave's B rows start at b_row0; CM half-tiles are [64 k][128 rows]: row index = column
        bf16x8 af[4][2], bf0[2][2], bf1[2][2];
        // phase 0: B cols 0..31, A rows 0..63 -> quadrant (0,0)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf0[j][ks] = fragB(cB, b_row0 + j * 16, ks);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = fragA(cA, i * 16, ks);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bf0[j][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        // phase 1: B cols 32..63 -> quadrant (0,1)
        if (SCHED == 1 && !ACM) {                                   // (a CM image of A is cut along k, not rows: nothing is free yet)
            __builtin_amdgcn_s_barrier();                           // every wave has read A rows 0-63: refill them for tile t+2
            asm volatile("" ::: "memory");
            if (more) { issue_piece(t + 2, buf, false, 0, 0); issue_piece(t + 2, buf, false, 1, 0); }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf1[j][ks] = fragB(cB, b_row0 + 32 + j * 16, ks);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bf1[j][ks], acc[i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        // phase 2: A rows 64..127 -> quadrant (1,1)
        if (SCHED == 1) {
            __builtin_amdgcn_s_barrier();                           // both B halves are in registers everywhere: refill them
            asm volatile("" ::: "memory");
            if (more) {
                issue_pieces(t + 2, buf, true, 0, 2, 0, 2);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = fragA(cA, 64 + i * 16, ks);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bf1[j][ks], acc[4 + i][2 + j], 0, 0, 0);
        // phase 3: quadrant (1,0) from registers
        if (DEEP) {
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();                           // the whole buffer has been consumed
            asm volatile("" ::: "memory");
            if (more) {
                if (SCHED == 1) {
                    issue_piece(t + 2, buf, false, 0, 1); issue_piece(t + 2, buf, false, 1, 1);
                    if (ACM) { issue_piece(t + 2, buf, false, 0, 0); issue_piece(t + 2, buf, false, 1, 0); }
                } else {
                    issue_tile(t + 2, buf);
                }
            }
            __builtin_amdgcn_s_setprio(1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bf0[j][ks], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    }
    __syncthreads();   // all LDS reads of the last tile done before the staging area is reused

    // ---- epilogue: 4 passes of 32 rows through per-wave fp32 staging; coalesced stores ----
    float* wst = (float*)smem + wave * 32 * T256_STG;
    constexpr int WN_OUT = GU ? 32 : 64;
    constexpr int NJ = GU ? 2 : 4;
    const int ncol0 = n0 + wc * WN_OUT;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 2 && MODE != 3 && MODE != 4 && MODE != 5) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = ncol0 + j * 16 + l15;
            bv[j] = (p.bias != nullptr && col < N) ? bf2f(p.bias[col]) : 0.f;
        }
    }
    constexpr int LPR = WN_OUT / 4, RPI = 64 / LPR;
    const int rr0 = lane / LPR, c4 = (lane % LPR) * 4;
    // MODE 3: one fp32 slab per K-slice; MODE 4: two planes (gate, up) per K-slice, ldc = columns of the tail region
    float* slab = (MODE == 3) ? (float*)p.C + (int64_t)blockIdx.y * M * p.ldc
                : (MODE == 4) ? (float*)p.C + (int64_t)blockIdx.y * 2 * M * p.ldc
                : (MODE == 5) ? (float*)p.C + ((int64_t)blockIdx.y * gridDim.x + (id - tile0)) * 65536 : nullptr;
    constexpr int PASSES = (MODE == 4) ? 2 : 1;
#pragma unroll
    for (int pl = 0; pl < PASSES; ++pl) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v;
                    if (MODE == 4) {
                        v = acc[2 * q + ii][j + 2 * pl][r];
                    } else if (GU) {
                        v = silu_f(acc[2 * q + ii][j][r]) * acc[2 * q + ii][j + 2][r];
                    } else {
                        v = acc[2 * q + ii][j][r] + bv[j];
                        if constexpr (EPI == EPI_GELU_TANH) v = gelu_tanh_f(v);
                        if constexpr (EPI == EPI_GELU_ERF) v = gelu_erf_f(v);
                    }
                    wst[(ii * 16 + lg * 4 + r) * T256_STG + j * 16 + l15] = v;
                }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int rr = it * RPI + rr0;
            const int gm = m0 + wr * (BM / 2) + q * 32 + rr, gc = ncol0 + c4;
            if (gm < M && gc < N) {
                f32x4 v = *(const f32x4*)(wst + rr * T256_STG + c4);
                if (MODE == 3) {
                    *(f32x4*)(slab + (int64_t)gm * p.ldc + gc) = v;
                } else if (MODE == 4) {
                    *(f32x4*)(slab + (int64_t)pl * M * p.ldc + (int64_t)gm * p.ldc + (gc - col0)) = v;
                } else if (MODE == 5) {
                    *(f32x4*)(slab + (gm - m0) * 256 + (gc - n0)) = v;
                } else {
                    if (p.residual != nullptr) {
                        const u32x2 rv = *(const u32x2*)(p.residual + (int64_t)(p.res_mod > 0 ? gm % p.res_mod : gm) * p.ldr + gc);
                        v[0] += lo_bf(rv[0]); v[1] += hi_bf(rv[0]); v[2] += lo_bf(rv[1]); v[3] += hi_bf(rv[1]);
                    }
                    if constexpr (MODE == 1) {
                        *(f32x4*)((float*)p.C + (int64_t)gm * p.ldc + gc) = v;
                    } else {
                        u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
                        *(u32x2*)((bf16_t*)p.C + (int64_t)gm * p.ldc + gc) = o;
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
    }
    // ---- EX: the extra fragment's 16 rows; this wave holds B fragments f0 = wr and f1 = wr + 2 of its 64 B rows ----
    if constexpr (EX) {
        if (has9) {
            const int f0 = wr, f1 = wr + 2;
            if constexpr (GU) {
                // gate fragment f0, up fragment f1 of the same 16 output columns (MODE 2: silu(g) * u; MODE 4: the two raw planes)
#pragma unroll
                for (int pl = 0; pl < PASSES; ++pl) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = (MODE == 4) ? acc9[pl][r] : silu_f(acc9[0][r]) * acc9[1][r];
                        wst[(lg * 4 + r) * T256_STG + l15] = v;
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                    const int rr = lane >> 2, c4 = (lane & 3) * 4;
                    const int gm = m0 + 256 + rr, gc = ncol0 + f0 * 16 + c4;
                    if (gm < M && gc < N) {
                        const f32x4 v = *(const f32x4*)(wst + rr * T256_STG + c4);
                        if (MODE == 4) {
                            *(f32x4*)(slab + (int64_t)pl * M * p.ldc + (int64_t)gm * p.ldc + (gc - col0)) = v;
                        } else {
                            u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
                            *(u32x2*)((bf16_t*)p.C + (int64_t)gm * p.ldc + gc) = o;
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int f = jj ? f1 : f0;
                    const float bvv = wr == 0 ? (jj ? bv[2] : bv[0]) : (jj ? bv[3] : bv[1]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc9[jj][r] + bvv;
                        if constexpr (EPI == EPI_GELU_TANH) v = gelu_tanh_f(v);
                        if constexpr (EPI == EPI_GELU_ERF) v = gelu_erf_f(v);
                        wst[(lg * 4 + r) * T256_STG + f * 16 + l15] = v;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int rr = it * 8 + (lane >> 3), g8 = lane & 7;
                    const int c4 = (g8 < 4 ? f0 : f1) * 16 + (g8 & 3) * 4;
                    const int gm = m0 + 256 + rr, gc = ncol0 + c4;
                    if (gm < M && gc < N) {
                        f32x4 v = *(const f32x4*)(wst + rr * T256_STG + c4);
                        if (MODE == 3) {
                            *(f32x4*)(slab + (int64_t)gm * p.ldc + gc) = v;
                        } else {
                            if (p.residual != nullptr) {
                                const u32x2 rv = *(const u32x2*)(p.residual + (int64_t)(p.res_mod > 0 ? gm % p.res_mod : gm) * p.ldr + gc);
                                v[0] += lo_bf(rv[0]); v[1] += hi_bf(rv[0]); v[2] += lo_bf(rv[1]); v[3] += hi_bf(rv[1]);
                            }
                            if constexpr (MODE == 1) {
                                *(f32x4*)((float*)p.C + (int64_t)gm * p.ldc + gc) = v;
                            } else {
                                u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
                                *(u32x2*)((bf16_t*)p.C + (int64_t)gm * p.ldc + gc) = o;
                            }
                        }
                    }
                }
            }
        }
    }
}

// 192-row tiles (BM) when they need fewer tile-times than 256-row tiles: rounds(tiles) x work per tile, a 192-row tile priced at 0.78 of a
// 256-row one (3/4 of the MFMAs on the same B traffic).  force: 0 = this rule, 192 / 256 = that tile height (vila_gemm_force_bm)
static inline bool prefer_bm192(int M, int N, int force) {
    if (force == 192) return true;
    if (force == 256) return false;
    const int t256 = cdiv(M, 256) * cdiv(N, 256), t192 = cdiv(M, 192) * cdiv(N, 256);
    return 0.78 * cdiv(t192, 256) < 0.97 * cdiv(t256, 256);
}

// rows the LAST 256-row tile carries as an extra 16-row fragment (EX kernels): M = 256 k + r, k >= 1, 1 <= r <= 16; else 0.
// VILA_GEMM_EX=0 switches the policy off (A/B measurements: the callers then see cdiv(M, 256) row tiles again)
extern int g_gemm256_ex;      // gemm256.hip: -1 = VILA_GEMM_EX from the environment (default 1), 0 = off, 1 = the policy below, 2 = whenever the rows fit (tests)
static inline int gemm256_ex_mode() {
    if (g_gemm256_ex < 0) { const char* e = getenv("VILA_GEMM_EX"); g_gemm256_ex = (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }
    return g_gemm256_ex;
}
static inline int gemm256_ex_rows(int M) {
    const int r = M % 256;
    return (gemm256_ex_mode() != 0 && M > 256 && r >= 1 && r <= 16) ? r : 0;
}
// 256-row tiles of an M-row output under that policy (what every launch policy must count with when it hands tile ranges to EX launches)
static inline int gemm256_tiles_m(int M) { return gemm256_ex_rows(M) ? M / 256 : cdiv(M, 256); }
// Whole-grid launches take the EX kernel only when dropping the extra row tile saves a ROUND of 256 blocks: its last-row blocks do 12.5 %
// more MFMAs and the kernel carries 16-28 more registers, measured 5-10 % slower than the plain kernel on grids with the same number of
// rounds (M = 3076 x N = 4608: 92 -> 101 us, profiles/r03_gemm_bench_fwd_ex.log); K-sliced launches always take it (fewer tiles = more slices)
static inline bool gemm256_ex_saves_round(int M, int tiles_n) {
    if (gemm256_ex_rows(M) == 0) return false;
    if (gemm256_ex_mode() == 2) return true;
    return cdiv((M / 256) * tiles_n, 256) < cdiv(cdiv(M, 256) * tiles_n, 256);
}

// tile range [tile0, tile0 + n_tiles) of the tile order (gemm256_tile_of; n_tiles < 0: all); per = K-tiles per slice for the split modes
template <int MODE, int EPI, bool ACM = false, bool BCM = false, int SCHED = 0, int BM = 256, bool EX = false>
static int launch256_t(const GemmArgs& a, hipStream_t s, int splits = 1, int tile0 = 0, int n_tiles = -1, int col0 = 0, int per = 0) {
    const int bn = (MODE == 2 || MODE == 4) ? 128 : 256;
    const int tiles_m = EX ? a.M / 256 : cdiv(a.M, BM), tiles_n = cdiv(a.N, bn);
    if (n_tiles < 0) n_tiles = tiles_m * tiles_n;
    const size_t lds = 2 * 4 * 128 * 64 * 2 + (EX ? 4096 : 0);   // 131072 >= 8 waves x 32 x 68 x 4 staging (+ the extra fragment's two 2-KB buffers)
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<MODE, EPI, ACM, BCM, SCHED, BM, EX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int kt = cdiv(a.K, T256_BK);
    if (per <= 0) per = kt / splits;
    // fused gate/up: grouped only when this launch covers the whole grid (its whole-rounds + sliced-tail policy cuts whole tile COLUMNS)
    const bool gu_partial = (MODE == 4) || (MODE == 2 && (tile0 != 0 || n_tiles != tiles_m * tiles_n));
    const int grp = gemm256_group(tiles_m, tiles_n, gu_partial);
    hipLaunchKernelGGL((gemm256_kernel<MODE, EPI, ACM, BCM, SCHED, BM, EX>), dim3(n_tiles, splits), dim3(512), lds, s, a, tiles_m, per, tile0, col0, grp);
    VILA_LAUNCH_CHECK();
    return 0;
}
// forward-layout launch that takes the EX kernel when the shape has 1..16 leftover rows (ex = gemm256_ex_rows(a.M) != 0, decided by the caller
// so that its tile ranges and this launch agree)
template <int MODE, int EPI>
static int launch256_fwd(const GemmArgs& a, hipStream_t s, bool ex, int splits = 1, int tile0 = 0, int n_tiles = -1, int col0 = 0, int per = 0) {
    if (ex) return launch256_t<MODE, EPI, false, false, T256_CC_SCHED, 256, true>(a, s, splits, tile0, n_tiles, col0, per);
    return launch256_t<MODE, EPI, false, false, T256_CC_SCHED>(a, s, splits, tile0, n_tiles, col0, per);
}
