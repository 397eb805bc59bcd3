// W8A8 GEMM for the vision tower (BASELINE configs[4]: "W8A8 vision tower"; the reference's numbers come from the
// external TinyChat backend, README.md:87 — nothing in-tree, so the format is defined here and in vila_amd/quant.py):
//   Y[M,N] = epi( (Xq[M,K] . Wq[N,K]^T) * sx[M] * sw[N] + bias[N] ) (+ residual)        Xq, Wq int8, accumulate int32 on the matrix cores
//   Wq / sw : per-OUTPUT-CHANNEL symmetric int8 (sw[n] = max|W[n,:]| / 127), quantised once (vila_amd/quant.py)
//   Xq / sx : per-TOKEN dynamic symmetric int8 (sx[m] = max|X[m,:]| / 127), produced by quant_rows_i8_kernel right before the GEMM
// Kernel = the 128x(32 NF) LDS-DMA ring of gemm_ring.hip with int8 operands: a K-tile is 128 int8 = the same 128-B LDS rows, the same
// 16-B slot swizzle, the same DMA addressing; v_mfma_i32_16x16x64_i8 consumes 16 B per lane and operand (2x the MAC rate of bf16).
// Both operands are fetched with the same lane -> k mapping, so the dot product does not depend on the instruction's internal k order.
#include "kernels.h"

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define I8_BM 128
#define I8_BKB 128      // K-tile in bytes (= int8 elements)

__device__ __attribute__((aligned(16))) unsigned int g_i8_zero_chunk[4];

struct GemmI8Args {
    const int8_t* A; int64_t lda;       // [M][K] int8
    const int8_t* W; int64_t ldw;       // [N][K] int8
    const float* sx;                    // [M]
    const float* sw;                    // [N]
    const bf16_t* bias;                 // [N] or null
    const bf16_t* residual; int64_t ldr;
    bf16_t* C; int64_t ldc;
    int M, N, K, epi;
};

template <int EPI, int STAGES, int NF>
__global__ __launch_bounds__(256, (STAGES * (I8_BM + 32 * NF) * I8_BKB <= 72 * 1024) ? 2 : 1) void gemm_i8_kernel(GemmI8Args p, int tiles_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN = 32 * NF;
    constexpr int A_BYTES = I8_BM * I8_BKB;              // 16 KB
    constexpr int B_BYTES = BN * I8_BKB;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int B_IT = BN / 32;
    constexpr int DMA_PER_TILE = 4 + B_IT;
    constexpr int STG = 16 * NF + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % tiles_m, tn = id / tiles_m;
    const int m0 = tm * I8_BM, n0 = tn * BN;
    const int M = p.M, N = p.N, K = p.K;

    uint32_t aoff[4], boff[B_IT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (i * 4 + wave) * 64 + lane, row = c >> 3;
        const int kch = (c & 7) ^ ((row >> 1) & 7);
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        aoff[i] = (uint32_t)gm * (uint32_t)p.lda + kch * 16;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int c = (i * 4 + wave) * 64 + lane, row = c >> 3;
        const int kch = (c & 7) ^ ((row >> 1) & 7);
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        boff[i] = (uint32_t)gn * (uint32_t)p.ldw + kch * 16;
    }
    const int kch_lane = (lane & 7) ^ (4 * (wave & 1) + (lane >> 4));
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);
    auto issue_tile = [&](int t) {
        const int k0 = t * I8_BKB;
        char* base = smem + (t % STAGES) * STAGE_BYTES + wave_lds;
        const bool kin = k0 + kch_lane * 16 < K;           // K % 16 == 0: a 16-B chunk is inside K or entirely beyond it
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int8_t* src = kin ? p.A + aoff[i] + k0 : (const int8_t*)g_i8_zero_chunk;
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(base + i * 4096), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int8_t* src = kin ? p.W + boff[i] + k0 : (const int8_t*)g_i8_zero_chunk;
            __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(base + A_BYTES + i * 4096), 16, 0, 0);
        }
    };

    const int swr = (l15 >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = l15 * 128 + (((ks * 4 + lg) ^ swr) << 4);

    i32x4 acc[4][NF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (i32x4){0, 0, 0, 0};

    const int nt = (K + I8_BKB - 1) / I8_BKB;
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nt) issue_tile(t);
    for (int t = 0; t < nt; ++t) {
        const int later = (nt - 1 - t) < (STAGES - 2) ? (nt - 1 - t) : (STAGES - 2);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_TILE) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + STAGES - 1 < nt) issue_tile(t + STAGES - 1);
        const char* cA = smem + (t % STAGES) * STAGE_BYTES + wr * 64 * 128;
        const char* cB = smem + (t % STAGES) * STAGE_BYTES + A_BYTES + wc * (16 * NF) * 128;
        i32x4 af[4][2], bfr[NF][2];
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bfr[j][ks] = *(const i32x4*)(cB + j * 16 * 128 + foff[ks]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const i32x4*)(cA + i * 16 * 128 + foff[ks]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[i][ks], bfr[j][ks], acc[i][j], 0, 0, 0);
    }
    __syncthreads();

    // ---- epilogue: acc * sw[col] staged per wave, then per row: * sx[row] + bias -> activation -> (+ residual) -> bf16 ----
    float* wst = (float*)smem + wave * 32 * STG;
    const int ncol0 = n0 + wc * (16 * NF);
    float swv[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int col = ncol0 + j * 16 + l15;
        swv[j] = col < N ? p.sw[col] : 0.f;
    }
    constexpr int LPR = 4 * NF, RPI = 64 / LPR;
    const int rr0 = lane / LPR, c4 = (lane % LPR) * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = (ncol0 + c4 + r < N) ? bf2f(p.bias[ncol0 + c4 + r]) : 0.f;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) wst[(ii * 16 + lg * 4 + r) * STG + j * 16 + l15] = (float)acc[2 * h + ii][j][r] * swv[j];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int rr = it * RPI + rr0;
            const int gm = m0 + wr * 64 + h * 32 + rr, gc = ncol0 + c4;
            if (gm < M && gc < N) {
                f32x4 v = *(const f32x4*)(wst + rr * STG + c4);
                const float sxm = p.sx[gm];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = v[r] * sxm + bv[r];
                    if constexpr (EPI == EPI_GELU_TANH) t = gelu_tanh_f(t);
                    if constexpr (EPI == EPI_GELU_ERF) t = gelu_erf_f(t);
                    v[r] = t;
                }
                if (p.residual != nullptr) {
                    const u32x2 rv = *(const u32x2*)(p.residual + (int64_t)gm * p.ldr + gc);
                    v[0] += lo_bf(rv[0]); v[1] += hi_bf(rv[0]); v[2] += lo_bf(rv[1]); v[3] += hi_bf(rv[1]);
                }
                u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
                *(u32x2*)(p.C + (int64_t)gm * p.ldc + gc) = o;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI, int STAGES, int NF>
static int launch_i8_t(const GemmI8Args& a, hipStream_t s) {
    const int tiles_m = cdiv(a.M, I8_BM), tiles_n = cdiv(a.N, 32 * NF);
    const size_t lds = (size_t)STAGES * (I8_BM + 32 * NF) * I8_BKB;
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)gemm_i8_kernel<EPI, STAGES, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_i8_kernel<EPI, STAGES, NF>), dim3(tiles_m * tiles_n), dim3(256), lds, s, a, tiles_m);
    VILA_LAUNCH_CHECK();
    return 0;
}

template <int STAGES, int NF>
static int launch_i8_epi(const GemmI8Args& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_NONE: return launch_i8_t<EPI_NONE, STAGES, NF>(a, s);
        case EPI_GELU_TANH: return launch_i8_t<EPI_GELU_TANH, STAGES, NF>(a, s);
    }
    VILA_FAIL(-1, "gemm_i8: unsupported epilogue %d", a.epi);
}

int launch_gemm_i8(const int8_t* A, int64_t lda, const int8_t* W, int64_t ldw, const float* sx, const float* sw, const bf16_t* bias,
                   const bf16_t* residual, int64_t ldr, bf16_t* C, int64_t ldc, int M, int N, int K, int epi, hipStream_t s) {
    VILA_REQUIRE(M > 0 && N > 0 && K >= I8_BKB, "gemm_i8: need M, N > 0 and K >= 128 (M=%d N=%d K=%d)", M, N, K);
    VILA_REQUIRE(K % 16 == 0 && N % 4 == 0 && lda % 16 == 0 && ldw % 16 == 0 && ldc % 4 == 0, "gemm_i8: K, lda, ldw must be multiples of 16, N and ldc of 4");
    VILA_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)W % 16 == 0) && ((uintptr_t)C % 8 == 0), "gemm_i8: pointers must be 16-B aligned");
    VILA_REQUIRE((int64_t)M * lda < (1ll << 31) && (int64_t)N * ldw < (1ll << 31), "gemm_i8: operand too large for 32-bit offsets");
    GemmI8Args a{A, lda, W, ldw, sx, sw, bias, residual, ldr, C, ldc, M, N, K, epi};
    const int64_t tiles128 = (int64_t)cdiv(M, 128) * cdiv(N, 128);
    if (tiles128 >= 270) return launch_i8_epi<2, 4>(a, s);      // 128x128 tile, 2 stages (as the bf16 ring picks)
    return launch_i8_epi<3, 2>(a, s);                           // 128x64 tile, 3 stages
}

// ------------------------------------------------------------------------------------------------------------------------------------
// per-token dynamic quantisation: x [rows][cols] bf16 -> q [rows][cols] int8, scale[rows] = max|x_row| / 127 (1 when the row is all zero)
// one wave per row (cols % 8 == 0): two passes over a row that stays in L1 / registers
// ------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_rows_i8_kernel(const bf16_t* __restrict__ x, int8_t* __restrict__ q, float* __restrict__ scale,
                                                            int rows, int cols) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + (int64_t)row * cols;
    const int c8 = cols >> 3;
    float amax = 0.f;
    for (int c = lane; c < c8; c += 64) {
        const u32x4 v = *(const u32x4*)(xr + c * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(lo_bf(v[j])), fabsf(hi_bf(v[j]))));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / 127.f : 1.f;        // q = round-half-even(x / sc): a true division, so the grid is the host rule's bit for bit
    if (lane == 0) scale[row] = sc;
    int8_t* qr = q + (int64_t)row * cols;
    for (int c = lane; c < c8; c += 64) {
        const u32x4 v = *(const u32x4*)(xr + c * 8);
        uint32_t o[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int a = (int)rintf(lo_bf(v[j]) / sc), b = (int)rintf(hi_bf(v[j]) / sc);
            a = a < -127 ? -127 : (a > 127 ? 127 : a); b = b < -127 ? -127 : (b > 127 ? 127 : b);
            o[j >> 1] |= ((uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8)) << ((j & 1) * 16);
        }
        *(u32x2*)(qr + c * 8) = (u32x2){o[0], o[1]};
    }
}
int launch_quant_rows_i8(const bf16_t* x, int8_t* q, float* scale, int rows, int cols, hipStream_t s) {
    VILA_REQUIRE(cols % 8 == 0 && rows > 0, "quant_rows_i8: cols (%d) must be a multiple of 8", cols);
    hipLaunchKernelGGL(quant_rows_i8_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, q, scale, rows, cols);
    VILA_LAUNCH_CHECK();
    return 0;
}
