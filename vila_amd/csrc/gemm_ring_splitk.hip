// K-SLICED variant of the 128x64 LDS-DMA ring GEMM (gemm_ring.hip) for SHORT PROMPTS: q/k/v/o of the LLM at M < 512 rows.
//
// Written at the end of round 4 without a GPU, measured in round 5 and ON by default since (gemm.hip launch_t; VILA_RING_SPLITK=0 turns it off,
// vila_gemm_force_tile(11) forces it): tools/gemm_bench prering, cold weights (profiles/r05_gemm_bench_prering.log) — qkv at M = 64 / 160 / 289
// 28.5 / 28.6 / 29.8 -> 17.8 / 20.8 / 29.5 us, o_proj + residual at M = 64 / 289 29.8 / 31.4 -> 14.2 / 24.3 us, Lite-3B qkv / o_proj at M = 154
// 18.0 / 19.0 -> 14.6 / 12.5 us; end to end the Lite-3B-shaped TTFT (S = 154) 8.67 -> 8.18 ms, NVILA-8B at S = 289 10.84 -> 10.82 ms
// (profiles/r05_second_call_ab.log).
//
// Why (profiles/r04_gemm_bench_presmall.log): at M = 64 .. 289 the q/k/v (N = 4608, K = 3584) and o_proj launches are
// 72 .. 216 blocks that each walk 56 K-tiles — 29-31 us whatever M is, i.e. the 33 / 26 MB of weights stream at ~1.1 TB/s, because the
// K loop of a lone block runs at memory latency (three tiles in flight) and under half of the 512 block slots are filled.  Cutting K over
// grid.y puts 2-4x the blocks (and loads) in flight; the partial sums go to fp32 slabs [slice][M][N] in the caller's workspace and a reduce
// pass adds bias / residual — the same contract as gemm256's split-K (gemm256.hip launch_gemm256_splitk), minus the fused follow-up norm
// (norm_done stays 0, so the caller launches its norm).  At M = 289, N = 4608 the slabs are 5.3 MB per slice: the reduce reads <= 21 MB.
//
// The kernel is the 3-stage 128x64 ring of gemm_ring.hip with (a) its K-tile range [t0, t1) taken from blockIdx.y and (b) an epilogue that
// stores the raw fp32 accumulators.  Tile / LDS layout, swizzle, counted waits: see gemm_ring.hip.
#include <stdlib.h>
#include "kernels.h"

#define RS_BM 128
#define RS_BN 64
#define RS_BK 64
#define RS_STAGES 3

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __attribute__((aligned(16))) unsigned int g_rsk_zero_chunk[4];   // K-tail source (zero-initialised)

__global__ __launch_bounds__(256, 2) void gemm_ring_splitk_kernel(GemmArgs p, int tiles_m, int per, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NF = 2;
    constexpr int A_BYTES = RS_BM * RS_BK * 2;           // 16 KB
    constexpr int B_BYTES = RS_BN * RS_BK * 2;           // 8 KB
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int B_IT = RS_BN / 32;
    constexpr int DMA_PER_TILE = 4 + B_IT;
    constexpr int STG = 16 * NF + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % tiles_m, tn = id / tiles_m;
    const int m0 = tm * RS_BM, n0 = tn * RS_BN;
    const int M = p.M, N = p.N, K = p.K;
    const int nt = (K + RS_BK - 1) / RS_BK;
    const int t0 = blockIdx.y * per;
    const int t1 = (t0 + per < nt) ? t0 + per : nt;      // the last slice takes what is left (host guarantees t0 < nt)

    uint32_t aoff[4], boff[B_IT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 4 + wave) * 64 + lane, row = c >> 3;
        const int kch = (c & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        aoff[i] = (uint32_t)gm * (uint32_t)p.lda + kch * 8;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int c = (i * 4 + wave) * 64 + lane, row = c >> 3;
        const int kch = (c & 7) ^ ((row >> 1) & 7);
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        boff[i] = (uint32_t)gn * (uint32_t)p.ldw + kch * 8;
    }
    const int kch_lane = (lane & 7) ^ (4 * (wave & 1) + (lane >> 4));
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);
    auto issue_tile = [&](int t) {                       // t = GLOBAL K-tile index; its ring stage is (t - t0) % RS_STAGES
        const int k0 = t * RS_BK;
        char* base = smem + ((t - t0) % RS_STAGES) * STAGE_BYTES + wave_lds;
        if (k0 + RS_BK <= K) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void*)(p.A + aoff[i] + k0), (lds_void*)(base + i * 4096), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                __builtin_amdgcn_global_load_lds((gbl_void*)(p.W + boff[i] + k0), (lds_void*)(base + A_BYTES + i * 4096), 16, 0, 0);
        } else {
            const bool kin = k0 + kch_lane * 8 < K;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16_t* src = kin ? p.A + aoff[i] + k0 : (const bf16_t*)g_rsk_zero_chunk;
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(base + i * 4096), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const bf16_t* src = kin ? p.W + boff[i] + k0 : (const bf16_t*)g_rsk_zero_chunk;
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(base + A_BYTES + i * 4096), 16, 0, 0);
            }
        }
    };

    const int swr = (l15 >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = l15 * 128 + (((ks * 4 + lg) ^ swr) << 4);

    f32x4 acc[4][NF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int i = 0; i < RS_STAGES - 1; ++i)
        if (t0 + i < t1) issue_tile(t0 + i);
    for (int t = t0; t < t1; ++t) {
        const int later = (t1 - 1 - t) < (RS_STAGES - 2) ? (t1 - 1 - t) : (RS_STAGES - 2);
        if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + RS_STAGES - 1 < t1) issue_tile(t + RS_STAGES - 1);
        const char* cA = smem + ((t - t0) % RS_STAGES) * STAGE_BYTES + wr * 64 * 128;
        const char* cB = smem + ((t - t0) % RS_STAGES) * STAGE_BYTES + A_BYTES + wc * (16 * NF) * 128;
        bf16x8 af[4][2], bfr[NF][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < NF; ++j) bfr[j][ks] = *(const bf16x8*)(cB + j * 16 * 128 + foff[ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i][ks] = *(const bf16x8*)(cA + i * 16 * 128 + foff[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bfr[j][ks], acc[i][j], 0, 0, 0);
    }
    __syncthreads();

    // ---- epilogue: the raw fp32 partial sums of this K-slice -> slab[blockIdx.y][M][N], two passes of 32 rows through per-wave staging ----
    float* wst = (float*)smem + wave * 32 * STG;
    float* out = slab + (int64_t)blockIdx.y * M * N;
    const int ncol0 = n0 + wc * (16 * NF);
    constexpr int LPR = 4 * NF, RPI = 64 / LPR;
    const int rr0 = lane / LPR, c4 = (lane % LPR) * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) wst[(ii * 16 + lg * 4 + r) * STG + j * 16 + l15] = acc[2 * h + ii][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int rr = it * RPI + rr0;
            const int gm = m0 + wr * 64 + h * 32 + rr, gc = ncol0 + c4;
            if (gm < M && gc < N) *(f32x4*)(out + (int64_t)gm * N + gc) = *(const f32x4*)(wst + rr * STG + c4);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// out[m][n] = bf16(sum_s slab[s][m][n] + bias[n] + residual[m][n])   (same arithmetic and order as gemm256.hip splitk_reduce_kernel)
__global__ void ring_splitk_reduce_kernel(const float* __restrict__ slab, int splits, int64_t slab_stride, const bf16_t* __restrict__ bias,
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

// How many K-slices the short-prompt policy would take for this problem (0 = do not slice): fill the 512 resident block slots (two
// 3-stage blocks per CU), at most 4 slices, at least 8 K-tiles per slice, slabs inside the workspace.
int gemm_ring_splitk_slices(const GemmArgs& a) {
    if (a.epi != EPI_NONE || a.out_f32 || a.ws == nullptr || a.a_cm || a.b_cm) return 0;
    if (a.K % 8 != 0 || a.N % 4 != 0 || (int64_t)a.M * a.lda >= (1ll << 31) || (int64_t)a.N * a.ldw >= (1ll << 31)) return 0;
    const int64_t tiles = (int64_t)cdiv(a.M, RS_BM) * cdiv(a.N, RS_BN);
    const int kt = cdiv(a.K, RS_BK);
    int splits = (int)(512 / tiles);
    if (splits > 4) splits = 4;
    while (splits >= 2 && (cdiv(kt, splits) < 8 || (size_t)splits * a.M * a.N * 4 > a.ws_bytes)) --splits;
    if (splits >= 2) splits = cdiv(kt, cdiv(kt, splits));      // drop empty trailing slices
    return splits >= 2 ? splits : 0;
}

int launch_gemm_ring_splitk(const GemmArgs& a, int splits, hipStream_t s) {
    const int kt = cdiv(a.K, RS_BK), per = cdiv(kt, splits);
    VILA_REQUIRE(a.epi == EPI_NONE && !a.out_f32 && a.ws != nullptr && splits >= 2 && (splits - 1) * per < kt &&
                 (size_t)splits * a.M * a.N * 4 <= a.ws_bytes, "gemm_ring split-K: %d K tiles / %d slices / workspace %zu B do not fit", kt, splits, a.ws_bytes);
    const int tiles_m = cdiv(a.M, RS_BM), tiles_n = cdiv(a.N, RS_BN);
    const size_t lds = (size_t)RS_STAGES * (RS_BM + RS_BN) * RS_BK * 2;
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)gemm_ring_splitk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_ring_splitk_kernel, dim3(tiles_m * tiles_n, splits), dim3(256), lds, s, a, tiles_m, per, a.ws);
    VILA_LAUNCH_CHECK();
    const int64_t total = (int64_t)a.M * (a.N / 4);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(ring_splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, a.ws, splits, (int64_t)a.M * a.N, a.bias, a.residual, a.ldr,
                       (bf16_t*)a.C, a.ldc, a.M, a.N, a.res_mod);
    VILA_LAUNCH_CHECK();
    return 0;
}
