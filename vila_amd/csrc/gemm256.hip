// 256x256x64 bf16 MFMA GEMM for the large-M contractions (SFT step: T = 3076 packed tokens; wgrad / dgrad; long prefill).
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
// Requires K % 64 == 0 (callers pad the contraction dim); M / N tails by row clamping + masked stores.
#include "kernels.h"

#define T256_BK 64
#define T256_STG 68

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <bool OUT_F32, int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmArgs p, int tiles_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int HALF_BYTES = 128 * 64 * 2;           // 16 KB
    constexpr int BUF_BYTES = 4 * HALF_BYTES;          // A_lo, A_hi, B_lo, B_hi
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % tiles_m, tn = id / tiles_m;
    const int m0 = tm * 256, n0 = tn * 256;
    const int M = p.M, N = p.N, K = p.K;

    // ---- DMA source offsets: thread's chunk c = tid + 512*i of a half-tile: row = c >> 3, LDS slot = c & 7 ----
    const int srow = tid >> 3;                         // + 64 i
    const int kch = (tid & 7) ^ ((srow >> 1) & 7);     // global k-chunk that lands in this lane's LDS slot
    uint32_t aoff[2][2], boff[2][2];                   // [half][round] element offsets of the row starts
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int gm = m0 + h * 128 + srow + 64 * i; gm = gm < M ? gm : M - 1;
            int gn = n0 + h * 128 + srow + 64 * i; gn = gn < N ? gn : N - 1;
            aoff[h][i] = (uint32_t)gm * (uint32_t)p.lda + kch * 8;
            boff[h][i] = (uint32_t)gn * (uint32_t)p.ldw + kch * 8;
        }
    const int lds_lane_base = __builtin_amdgcn_readfirstlane(wave * 1024);   // wave-uniform: 64 lanes x 16 B per DMA
    auto issue_tile = [&](int t, int buf) {
        const int k0 = t * T256_BK;
        char* base = smem + buf * BUF_BYTES + lds_lane_base;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                __builtin_amdgcn_global_load_lds((gbl_void*)(p.A + aoff[h][i] + k0), (lds_void*)(base + h * HALF_BYTES + i * 8192), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gbl_void*)(p.W + boff[h][i] + k0), (lds_void*)(base + (2 + h) * HALF_BYTES + i * 8192), 16, 0, 0);
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

    const int nt = K / T256_BK;
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

    // ---- epilogue: 4 passes of 32 rows through per-wave fp32 staging; bias / GELU / residual; coalesced stores ----
    float* wst = (float*)smem + wave * 32 * T256_STG;
    const int ncol0 = n0 + wc * 64;
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = ncol0 + j * 16 + l15;
        bv[j] = (p.bias != nullptr && col < N) ? bf2f(p.bias[col]) : 0.f;
    }
    const int rr0 = lane >> 4, c4 = (lane & 15) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[2 * q + ii][j][r] + bv[j];
                    if constexpr (EPI == EPI_GELU_TANH) v = gelu_tanh_f(v);
                    if constexpr (EPI == EPI_GELU_ERF) v = gelu_erf_f(v);
                    wst[(ii * 16 + lg * 4 + r) * T256_STG + j * 16 + l15] = v;
                }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int rr = it * 4 + rr0;
            const int gm = m0 + wr * 128 + q * 32 + rr, gc = ncol0 + c4;
            if (gm < M && gc < N) {
                f32x4 v = *(const f32x4*)(wst + rr * T256_STG + c4);
                if (p.residual != nullptr) {
                    const u32x2 rv = *(const u32x2*)(p.residual + (int64_t)gm * p.ldr + gc);
                    v[0] += lo_bf(rv[0]); v[1] += hi_bf(rv[0]); v[2] += lo_bf(rv[1]); v[3] += hi_bf(rv[1]);
                }
                if constexpr (OUT_F32) {
                    *(f32x4*)((float*)p.C + (int64_t)gm * p.ldc + gc) = v;
                } else {
                    u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
                    *(u32x2*)((bf16_t*)p.C + (int64_t)gm * p.ldc + gc) = o;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool OUT_F32, int EPI>
static int launch256_t(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = cdiv(a.M, 256), tiles_n = cdiv(a.N, 256);
    const size_t lds = 2 * 4 * 128 * 64 * 2;   // 131072 >= 8 waves x 32 x 68 x 4 staging
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<OUT_F32, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm256_kernel<OUT_F32, EPI>), dim3(tiles_m * tiles_n), dim3(512), lds, s, a, tiles_m);
    VILA_LAUNCH_CHECK();
    return 0;
}

bool gemm256_supported(const GemmArgs& a) {
    return a.epi != EPI_GATEUP && a.K % T256_BK == 0 && a.K >= 2 * T256_BK &&
           (int64_t)a.M * a.lda < (1ll << 31) && (int64_t)a.N * a.ldw < (1ll << 31);
}

int launch_gemm256(const GemmArgs& a, hipStream_t s) {
    if (a.out_f32) return launch256_t<true, EPI_NONE>(a, s);
    switch (a.epi) {
        case EPI_NONE: return launch256_t<false, EPI_NONE>(a, s);
        case EPI_GELU_TANH: return launch256_t<false, EPI_GELU_TANH>(a, s);
        case EPI_GELU_ERF: return launch256_t<false, EPI_GELU_ERF>(a, s);
    }
    VILA_FAIL(-1, "gemm256: unsupported epilogue %d", a.epi);
}
