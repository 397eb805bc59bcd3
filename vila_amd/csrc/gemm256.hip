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
#include "kernels.h"

#define T256_BK 64
#define T256_STG 68

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __attribute__((aligned(16))) unsigned int g_zero_chunk[4];   // K-tail source (zero-initialised)

// MODE: 0 = bf16 out (bias/GELU/residual), 1 = fp32 out, 2 = gate/up fused (bf16 out), 3 = split-K fp32 slab (raw accumulators),
//       4 = gate/up split-K: raw gate and up accumulators into two fp32 planes per K-slice (the tail round of an under-filled grid)
// tile0 = first tile id of this launch (a GEMM may be issued as "full rounds" + "split tail"), col0 = first output column of the slab
template <int MODE, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p, int tiles_m, int k_tiles_per_split, int tile0, int col0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HALF_BYTES = 128 * 64 * 2;           // 16 KB
    constexpr int BUF_BYTES = 4 * HALF_BYTES;          // A_lo, A_hi, B_lo, B_hi
    constexpr bool GU = (MODE == 2 || MODE == 4);
    constexpr bool SPLIT = (MODE == 3 || MODE == 4);
    constexpr int BN_OUT = GU ? 128 : 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, gridDim.x) + tile0;
    const int tm = id % tiles_m, tn = id / tiles_m;
    const int m0 = tm * 256, n0 = tn * BN_OUT;
    const int M = p.M, N = p.N;

    // ---- DMA source offsets: thread's chunk c = tid + 512*i of a half-tile: row = c >> 3, LDS slot = c & 7 ----
    const int srow = tid >> 3;                         // + 64 i
    const int kch = (tid & 7) ^ ((srow >> 1) & 7);     // global k-chunk that lands in this lane's LDS slot
    uint32_t aoff[2][2], boff[2][2];                   // [half][round] element offsets of the row starts
    bool b_up[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int gm = m0 + h * 128 + srow + 64 * i; gm = gm < M ? gm : M - 1;
            aoff[h][i] = (uint32_t)gm * (uint32_t)p.lda + kch * 8;
            const int rb = h * 128 + srow + 64 * i;    // row of the 256-row B tile; wave column = rb >> 6
            int gn;
            if (GU) { gn = n0 + (rb >> 6) * 32 + (rb & 31); b_up[h][i] = (rb & 32) != 0; }
            else { gn = n0 + rb; b_up[h][i] = false; }
            gn = gn < N ? gn : N - 1;
            boff[h][i] = (uint32_t)gn * (uint32_t)p.ldw + kch * 8;
        }
    const int lds_lane_base = __builtin_amdgcn_readfirstlane(wave * 1024);   // wave-uniform: 64 lanes x 16 B per DMA
    const int kt0 = SPLIT ? blockIdx.y * k_tiles_per_split : 0;
    auto issue_tile = [&](int t, int buf) {
        const int k0 = (kt0 + t) * T256_BK;
        char* base = smem + buf * BUF_BYTES + lds_lane_base;
        if (k0 + T256_BK <= p.K) {                          // block-uniform: every K-tile but a ragged last one
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    __builtin_amdgcn_global_load_lds((gbl_void*)(p.A + aoff[h][i] + k0), (lds_void*)(base + h * HALF_BYTES + i * 8192), 16, 0, 0);
                    const bf16_t* wsrc = (GU && b_up[h][i]) ? p.W2 : p.W;
                    __builtin_amdgcn_global_load_lds((gbl_void*)(wsrc + boff[h][i] + k0), (lds_void*)(base + (2 + h) * HALF_BYTES + i * 8192), 16, 0, 0);
                }
        } else {
            const bool kin = k0 + kch * 8 < p.K;            // this lane's 16-B chunk is inside K
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16_t* asrc = kin ? p.A + aoff[h][i] + k0 : (const bf16_t*)g_zero_chunk;
                    __builtin_amdgcn_global_load_lds((gbl_void*)asrc, (lds_void*)(base + h * HALF_BYTES + i * 8192), 16, 0, 0);
                    const bf16_t* wsrc = (GU && b_up[h][i]) ? p.W2 : p.W;
                    wsrc = kin ? wsrc + boff[h][i] + k0 : (const bf16_t*)g_zero_chunk;
                    __builtin_amdgcn_global_load_lds((gbl_void*)wsrc, (lds_void*)(base + (2 + h) * HALF_BYTES + i * 8192), 16, 0, 0);
                }
        }
    };

    // ---- fragment read offsets (bytes inside a half-tile) ----
    const int swr = (l15 >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = l15 * 128 + (((ks * 4 + lg) ^ swr) << 4);
    const int a_half = wr, b_half = wc >> 1, b_row0 = (wc & 1) * 64;

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int kt_all = (p.K + T256_BK - 1) / T256_BK;
    const int nt = SPLIT ? ((kt_all - kt0) < k_tiles_per_split ? (kt_all - kt0) : k_tiles_per_split) : kt_all;   // last slice may be shorter
    issue_tile(0, 0);
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own DMA of tile t has landed
        __builtin_amdgcn_s_barrier();                               // everyone's has; everyone left tile t-1
        asm volatile("" ::: "memory");                              // keep the LDS reads of tile t below the barrier
        if (t + 1 < nt) issue_tile(t + 1, buf ^ 1);
        const char* cA = smem + buf * BUF_BYTES + a_half * HALF_BYTES;
        const char* cB = smem + buf * BUF_BYTES + (2 + b_half) * HALF_BYTES + b_row0 * 128;
        bf16x8 af[4][2], bf0[2][2], bf1[2][2];
        // phase 0: B cols 0..31, A rows 0..63 -> quadrant (0,0)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf0[j][ks] = *(const bf16x8*)(cB + j * 16 * 128 + foff[ks]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const bf16x8*)(cA + i * 16 * 128 + foff[ks]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bf0[j][ks], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        // phase 1: B cols 32..63 -> quadrant (0,1)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf1[j][ks] = *(const bf16x8*)(cB + (32 + j * 16) * 128 + foff[ks]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bf1[j][ks], acc[i][2 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        // phase 2: A rows 64..127 -> quadrant (1,1)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const bf16x8*)(cA + (64 + i * 16) * 128 + foff[ks]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bf1[j][ks], acc[4 + i][2 + j], 0, 0, 0);
        // phase 3: quadrant (1,0) from registers
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bf0[j][ks], acc[4 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();   // all LDS reads of the last tile done before the staging area is reused

    // ---- epilogue: 4 passes of 32 rows through per-wave fp32 staging; coalesced stores ----
    float* wst = (float*)smem + wave * 32 * T256_STG;
    constexpr int WN_OUT = GU ? 32 : 64;
    constexpr int NJ = GU ? 2 : 4;
    const int ncol0 = n0 + wc * WN_OUT;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 2 && MODE != 3 && MODE != 4) {
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
                : (MODE == 4) ? (float*)p.C + (int64_t)blockIdx.y * 2 * M * p.ldc : nullptr;
    constexpr int PASSES = (MODE == 4) ? 2 : 1;
#pragma unroll
    for (int pl = 0; pl < PASSES; ++pl) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
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
            const int gm = m0 + wr * 128 + q * 32 + rr, gc = ncol0 + c4;
            if (gm < M && gc < N) {
                f32x4 v = *(const f32x4*)(wst + rr * T256_STG + c4);
                if (MODE == 3) {
                    *(f32x4*)(slab + (int64_t)gm * p.ldc + gc) = v;
                } else if (MODE == 4) {
                    *(f32x4*)(slab + (int64_t)pl * M * p.ldc + (int64_t)gm * p.ldc + (gc - col0)) = v;
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
}

// tail round of a gate/up GEMM: out[m][col0 + c] = bf16(silu(sum_s gate_s[m][c]) * sum_s up_s[m][c]); slab = [splits][2][M][tc] fp32
__global__ void splitk_gu_reduce_kernel(const float* __restrict__ slab, int splits, bf16_t* __restrict__ out, int64_t ldc, int M, int tc, int col0) {
    const int n4 = tc >> 2;
    const int64_t total = (int64_t)M * n4, plane = (int64_t)M * tc;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), c = (int)(i % n4) * 4;
        f32x4 g = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
        for (int sidx = 0; sidx < splits; ++sidx) {
            const float* b = slab + (int64_t)sidx * 2 * plane + (int64_t)m * tc + c;
            const f32x4 a = *(const f32x4*)b, w = *(const f32x4*)(b + plane);
            g[0] += a[0]; g[1] += a[1]; g[2] += a[2]; g[3] += a[3];
            u[0] += w[0]; u[1] += w[1]; u[2] += w[2]; u[3] += w[3];
        }
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
        f32x4 v = *(const f32x4*)(slab + (int64_t)m * N + c);
        for (int s = 1; s < splits; ++s) {
            const f32x4 w = *(const f32x4*)(slab + s * slab_stride + (int64_t)m * N + c);
            v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
        }
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

// tile range [tile0, tile0 + n_tiles) of the tm-fastest tile order (n_tiles < 0: all); per = K-tiles per slice for the split modes
template <int MODE, int EPI>
static int launch256_t(const GemmArgs& a, hipStream_t s, int splits = 1, int tile0 = 0, int n_tiles = -1, int col0 = 0, int per = 0) {
    const int bn = (MODE == 2 || MODE == 4) ? 128 : 256;
    const int tiles_m = cdiv(a.M, 256), tiles_n = cdiv(a.N, bn);
    if (n_tiles < 0) n_tiles = tiles_m * tiles_n;
    const size_t lds = 2 * 4 * 128 * 64 * 2;   // 131072 >= 8 waves x 32 x 68 x 4 staging
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<MODE, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int kt = cdiv(a.K, T256_BK);
    if (per <= 0) per = kt / splits;
    hipLaunchKernelGGL((gemm256_kernel<MODE, EPI>), dim3(n_tiles, splits), dim3(512), lds, s, a, tiles_m, per, tile0, col0);
    VILA_LAUNCH_CHECK();
    return 0;
}

bool gemm256_supported(const GemmArgs& a) {
    return a.K % 8 == 0 && a.K >= 2 * T256_BK && (int64_t)a.M * a.lda < (1ll << 31) && (int64_t)a.N * a.ldw < (1ll << 31);
}

// Gate/up with an under-filled LAST round (S = 769: 592 tiles = 2 full rounds of 256 + 80): the full rounds run fused as usual, the
// tail tiles are sliced over K so the last round costs 1/splits of a tile time; raw gate / up sums meet in a small reduce kernel.
static int launch_gateup(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = cdiv(a.M, 256), tiles_n = cdiv(a.N, 128), kt = cdiv(a.K, T256_BK);
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
            VILA_TRY((launch256_t<2, EPI_NONE>(a, s, 1, 0, full_tn * tiles_m)));
            GemmArgs b = a;
            b.C = a.ws; b.ldc = tc;
            VILA_TRY((launch256_t<4, EPI_NONE>(b, s, splits, full_tn * tiles_m, tail_tiles, full_tn * 128, per)));
            const int64_t total = (int64_t)a.M * (tc / 4);
            const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
            hipLaunchKernelGGL(splitk_gu_reduce_kernel, dim3(grid), dim3(256), 0, s, a.ws, splits, (bf16_t*)a.C, a.ldc, a.M, tc, full_tn * 128);
            VILA_LAUNCH_CHECK();
            return 0;
        }
    }
    return launch256_t<2, EPI_NONE>(a, s);
}

int launch_gemm256(const GemmArgs& a, hipStream_t s) {
    if (a.epi == EPI_GATEUP) return launch_gateup(a, s);
    if (a.out_f32) return launch256_t<1, EPI_NONE>(a, s);
    switch (a.epi) {
        case EPI_NONE: return launch256_t<0, EPI_NONE>(a, s);
        case EPI_GELU_TANH: return launch256_t<0, EPI_GELU_TANH>(a, s);
        case EPI_GELU_ERF: return launch256_t<0, EPI_GELU_ERF>(a, s);
    }
    VILA_FAIL(-1, "gemm256: unsupported epilogue %d", a.epi);
}

// split-K: C = sum over `splits` K-slices; `slab` = splits * M * N fp32 workspace owned by the caller
int launch_gemm256_splitk(const GemmArgs& a, int splits, float* slab, hipStream_t s) {
    const int kt = cdiv(a.K, T256_BK), per = cdiv(kt, splits);
    VILA_REQUIRE(a.epi == EPI_NONE && !a.out_f32 && a.N % 4 == 0 && splits >= 1 && (splits - 1) * per < kt,
                 "gemm256 split-K: %d K tiles cannot be cut into %d non-empty slices", kt, splits);
    GemmArgs b = a;
    b.C = slab; b.ldc = a.N; b.bias = nullptr; b.residual = nullptr;
    VILA_TRY((launch256_t<3, EPI_NONE>(b, s, splits, 0, -1, 0, per)));      // the last slice takes the remainder
    const int64_t total = (int64_t)a.M * (a.N / 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, slab, splits, (int64_t)a.M * a.N, a.bias, a.residual, a.ldr,
                       (bf16_t*)a.C, a.ldc, a.M, a.N, a.res_mod);
    VILA_LAUNCH_CHECK();
    return 0;
}
