// bf16 MFMA GEMM for the prefill / ViT / projector / lm_head contractions (SURVEY.md §8 rows a2,a3,a5,a10,a11).
//
//   C[M,N] = epi(A[M,K] . W[N,K]^T + bias) (+ residual)        nn.Linear layout: W is [out,in] row-major,
//   so BOTH operands are K-contiguous = the natural MFMA A/B fragment layout (8 consecutive k per lane).
//
// Tile 128x128x64, 256 threads = 4 waves (2x2), wave tile 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 fragments.
// HBM -> registers -> LDS (XOR-swizzled 16-B slots, double buffered, one barrier per K-tile; loads for tile t+1
// are issued before the MFMAs of tile t and written to LDS after them).  Epilogue goes through LDS so that the
// residual read and the C write are full-row coalesced (8 / 16 B per lane).
// Roofline: MFMA-bound at M >= 256 (2*M*N*K flop vs (M+N)*K*2 + M*N*2 bytes).
#include "kernels.h"

#define BM 128
#define BK 64
#define STG 68  // fp32 staging row stride (floats)

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_bf16_tn(GemmArgs p, int tiles_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sA = (bf16_t*)smem;         // [2][128][64]
    bf16_t* sB = sA + 2 * BM * BK;      // [2][128][64]
    float* stage = (float*)smem;        // epilogue reuse: [4][64][STG]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % tiles_m, tn = id / tiles_m;
    constexpr int BN_OUT = (EPI == EPI_GATEUP) ? 64 : 128;
    const int m0 = tm * BM, n0 = tn * BN_OUT;
    const int M = p.M, N = p.N, K = p.K;

    // ---- staging coordinates: thread owns 16-B chunk (row = r0 + 32 i, kc) of both tiles ----
    const int kc = tid & 7, r0 = tid >> 3;
    const int sw = kc ^ ((r0 >> 1) & 7);
    const bf16_t* a_ptr[4];
    const bf16_t* b_ptr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r0 + 32 * i;
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        a_ptr[i] = p.A + (int64_t)gm * p.lda + kc * 8;
        if constexpr (EPI == EPI_GATEUP) {
            const int j = (row >> 4) & 3;
            int gn = n0 + (row >> 6) * 32 + (j & 1) * 16 + (row & 15); gn = gn < N ? gn : N - 1;
            b_ptr[i] = ((j >> 1) ? p.W2 : p.W) + (int64_t)gn * p.ldw + kc * 8;
        } else {
            int gn = n0 + row; gn = gn < N ? gn : N - 1;
            b_ptr[i] = p.W + (int64_t)gn * p.ldw + kc * 8;
        }
    }
    const int st_off = r0 * BK + sw * 8;  // + 32*i*BK per chunk

    // ---- fragment read offsets (elements) ----
    const int sw_r = (l15 >> 1) & 7;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int kcc = ks * 4 + lg;
        a_off[ks] = (wr * 64 + l15) * BK + ((kcc ^ sw_r) << 3);
        b_off[ks] = (wc * 64 + l15) * BK + ((kcc ^ sw_r) << 3);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = (K + BK - 1) / BK;
    u32x4 ra[4], rb[4];
    auto gload = [&](int t) {
        const int k0 = t * BK;
        const bool ok = (k0 + kc * 8) < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = ok ? *(const u32x4*)(a_ptr[i] + k0) : (u32x4){0u, 0u, 0u, 0u};
            rb[i] = ok ? *(const u32x4*)(b_ptr[i] + k0) : (u32x4){0u, 0u, 0u, 0u};
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(sA + buf * BM * BK + st_off + 32 * i * BK) = ra[i];
            *(u32x4*)(sB + buf * BM * BK + st_off + 32 * i * BK) = rb[i];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload(t + 1);
        const bf16_t* cA = sA + buf * BM * BK;
        const bf16_t* cB = sB + buf * BM * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(cA + a_off[ks] + i * 16 * BK);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *(const bf16x8*)(cB + b_off[ks] + j * 16 * BK);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias / activation in registers -> per-wave fp32 staging -> coalesced residual add + store ----
    float* wst = stage + wave * 64 * STG;
    constexpr int WN = (EPI == EPI_GATEUP) ? 32 : 64;  // output columns per wave
    const int ncol0 = n0 + wc * WN;
    if constexpr (EPI == EPI_GATEUP) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = acc[i][j][r], u = acc[i][j + 2][r];
                    wst[(i * 16 + lg * 4 + r) * STG + j * 16 + l15] = silu_f(g) * u;
                }
    } else {
        float bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = ncol0 + j * 16 + l15;
            bv[j] = (p.bias != nullptr && col < N) ? bf2f(p.bias[col]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r] + bv[j];
                    if constexpr (EPI == EPI_GELU_TANH) v = gelu_tanh_f(v);
                    if constexpr (EPI == EPI_GELU_ERF) v = gelu_erf_f(v);
                    wst[(i * 16 + lg * 4 + r) * STG + j * 16 + l15] = v;
                }
    }
    // each wave only reads back what it wrote itself: a wave-level LDS fence is enough
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();

    constexpr int LPR = WN / 4;          // lanes per output row
    constexpr int RPI = 64 / LPR;        // rows per pass
    const int rr0 = lane / LPR, c4 = (lane % LPR) * 4;
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
        const int rr = it * RPI + rr0;
        const int gm = m0 + wr * 64 + rr, gc = ncol0 + c4;
        if (gm < M && gc < N) {
            f32x4 v = *(const f32x4*)(wst + rr * STG + c4);
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
}

template <int EPI, bool OUT_F32>
static int launch_t(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = cdiv(a.M, BM);
    const int bn = (EPI == EPI_GATEUP) ? 64 : 128;
    const int tiles_n = cdiv(a.N, bn);
    const size_t lds = 4 * 64 * STG * sizeof(float);  // 69632 >= 2*2*128*64*2
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)gemm_bf16_tn<EPI, OUT_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_tn<EPI, OUT_F32>), dim3(tiles_m * tiles_n), dim3(256), lds, s, a, tiles_m);
    VILA_LAUNCH_CHECK();
    return 0;
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    VILA_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    VILA_REQUIRE(a.K % 8 == 0 && a.N % 4 == 0, "gemm: K (%d) must be a multiple of 8 and N (%d) of 4", a.K, a.N);
    VILA_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldc % 4 == 0, "gemm: leading dims must keep 16-B row alignment");
    VILA_REQUIRE(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.W % 16 == 0) && ((uintptr_t)a.C % 16 == 0), "gemm: pointers must be 16-B aligned");
    VILA_REQUIRE(a.residual == nullptr || (a.ldr % 4 == 0 && (uintptr_t)a.residual % 8 == 0), "gemm: residual alignment");
    if (a.epi == EPI_GATEUP) {
        VILA_REQUIRE(a.W2 != nullptr && !a.out_f32 && a.bias == nullptr, "gemm: gate/up mode needs W2, bf16 out, no bias");
        VILA_REQUIRE((uintptr_t)a.W2 % 16 == 0, "gemm: W2 alignment");
        return launch_t<EPI_GATEUP, false>(a, s);
    }
    if (a.out_f32) {
        VILA_REQUIRE(a.epi == EPI_NONE, "gemm: fp32 output only with EPI_NONE");
        return launch_t<EPI_NONE, true>(a, s);
    }
    switch (a.epi) {
        case EPI_NONE: return launch_t<EPI_NONE, false>(a, s);
        case EPI_GELU_TANH: return launch_t<EPI_GELU_TANH, false>(a, s);
        case EPI_GELU_ERF: return launch_t<EPI_GELU_ERF, false>(a, s);
    }
    VILA_FAIL(-1, "gemm: unknown epilogue %d", a.epi);
}
