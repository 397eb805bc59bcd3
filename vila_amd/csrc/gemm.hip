// bf16 MFMA GEMM for the prefill / ViT / projector / lm_head contractions and the dgrad / wgrad GEMMs of the SFT step
// (SURVEY.md §8 rows a2,a3,a5,a10,a11,a13).
//
//   C[M,N] = epi(A[M,K] . W[N,K]^T + bias) (+ residual)        nn.Linear layout: W is [out,in] row-major,
//   so BOTH operands are K-contiguous = the natural MFMA A/B fragment layout (8 consecutive k per lane).
//
// One kernel template, three tile shapes (waves are WM x 2, each wave owns 64 x (16*NF) of C as 4 x NF
// v_mfma_f32_16x16x32_bf16 fragments, K-tile 64):
//     256 x 128  (8 waves)  M >= 1536: SFT-step shapes; halves the B-tile bytes per flop (the 128^2 tile is exactly
//                           L1/LDS-bound on gfx950: 32 KB per K-tile at 64 B/clk == its 512 MFMA cycles)
//     128 x 128  (4 waves)  default
//     128 x  64  (4 waves)  small grids (S = 769 prefill with N <= 4608: 196 tiles of 128^2 cannot fill 256 CUs)
// HBM -> registers -> LDS (XOR-swizzled 16-B slots, double buffered, one barrier per K-tile; loads for tile t+1 are
// issued before the MFMAs of tile t and written to LDS after them).  The epilogue goes through LDS so that the
// residual read and the C write are full-row coalesced (8 / 16 B per lane).
// Roofline: MFMA (2*M*N*K flop vs (M+N)*K*2 + M*N*2 bytes).
#include "kernels.h"

#define BK 64
#define STG 68  // fp32 staging row stride (floats)

template <int EPI, bool OUT_F32, int WM, int NF>
__global__ __launch_bounds__(WM * 128, (WM == 2) ? 2 : 2) void gemm_bf16_tn(GemmArgs p, int tiles_m) {
    constexpr int THREADS = WM * 128;
    constexpr int BM = WM * 64;              // rows of C per block
    constexpr int BNT = 32 * NF;             // B-tile rows held in LDS (2 waves x 16*NF)
    constexpr int A_IT = BM * 8 / THREADS;   // 16-B chunks per thread and K-tile
    constexpr int B_IT = (BNT * 8 + THREADS - 1) / THREADS;
    constexpr int RSTEP = THREADS / 8;       // rows covered per staging pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sA = (bf16_t*)smem;              // [2][BM][64]
    bf16_t* sB = sA + 2 * BM * BK;           // [2][BNT][64]
    float* stage = (float*)smem;             // epilogue reuse: [waves][64][STG]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % tiles_m, tn = id / tiles_m;
    constexpr int WN = 16 * NF;                                   // C columns per wave held in registers
    constexpr int WN_OUT = (EPI == EPI_GATEUP) ? WN / 2 : WN;     // C columns per wave written
    constexpr int BN_OUT = 2 * WN_OUT;
    const int m0 = tm * BM, n0 = tn * BN_OUT;
    const int M = p.M, N = p.N, K = p.K;

    // ---- staging coordinates: thread owns 16-B chunk (row = r0 + RSTEP*i, kc) ----
    const int kc = tid & 7, r0 = tid >> 3;
    const int sw = kc ^ ((r0 >> 1) & 7);     // RSTEP is a multiple of 16 => same swizzle for every pass
    const bf16_t* a_ptr[A_IT];
    const bf16_t* b_ptr[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int gm = m0 + r0 + RSTEP * i; gm = gm < M ? gm : M - 1;
        a_ptr[i] = p.A + (int64_t)gm * p.lda + kc * 8;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = r0 + RSTEP * i;      // B-tile row: wave column = row / WN, fragment j = (row % WN) / 16
        if constexpr (EPI == EPI_GATEUP) {
            const int j = (row % WN) >> 4;
            int gn = n0 + (row / WN) * WN_OUT + (j % (NF / 2)) * 16 + (row & 15); gn = gn < N ? gn : N - 1;
            b_ptr[i] = ((j >= NF / 2) ? p.W2 : p.W) + (int64_t)gn * p.ldw + kc * 8;
        } else {
            int gn = n0 + row; gn = gn < N ? gn : N - 1;
            b_ptr[i] = p.W + (int64_t)gn * p.ldw + kc * 8;
        }
    }
    const int st_off = r0 * BK + sw * 8;
    const bool b_active = (B_IT * RSTEP == BNT) || (r0 < BNT);   // 128x64 tile with 256 threads: only half the threads stage B

    // ---- fragment read offsets (elements) ----
    const int sw_r = (l15 >> 1) & 7;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int kcc = ks * 4 + lg;
        a_off[ks] = (wr * 64 + l15) * BK + ((kcc ^ sw_r) << 3);
        b_off[ks] = (wc * WN + l15) * BK + ((kcc ^ sw_r) << 3);
    }

    f32x4 acc[4][NF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = (K + BK - 1) / BK;
    u32x4 ra[A_IT], rb[B_IT];
    auto gload = [&](int t) {
        const int k0 = t * BK;
        const bool ok = (k0 + kc * 8) < K;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[i] = ok ? *(const u32x4*)(a_ptr[i] + k0) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[i] = (ok && b_active) ? *(const u32x4*)(b_ptr[i] + k0) : (u32x4){0u, 0u, 0u, 0u};
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *(u32x4*)(sA + buf * BM * BK + st_off + RSTEP * i * BK) = ra[i];
        if (b_active) {
#pragma unroll
            for (int i = 0; i < B_IT; ++i) *(u32x4*)(sB + buf * BNT * BK + st_off + RSTEP * i * BK) = rb[i];
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload(t + 1);
        const bf16_t* cA = sA + buf * BM * BK;
        const bf16_t* cB = sB + buf * BNT * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], bfr[NF];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8*)(cA + a_off[ks] + i * 16 * BK);
#pragma unroll
            for (int j = 0; j < NF; ++j) bfr[j] = *(const bf16x8*)(cB + b_off[ks] + j * 16 * BK);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias / activation in registers -> per-wave fp32 staging -> coalesced residual add + store ----
    float* wst = stage + wave * 64 * STG;
    const int ncol0 = n0 + wc * WN_OUT;
    if constexpr (EPI == EPI_GATEUP) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NF / 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float g = acc[i][j][r], u = acc[i][j + NF / 2][r];
                    wst[(i * 16 + lg * 4 + r) * STG + j * 16 + l15] = silu_f(g) * u;
                }
    } else {
        float bv[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int col = ncol0 + j * 16 + l15;
            bv[j] = (p.bias != nullptr && col < N) ? bf2f(p.bias[col]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
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

    constexpr int LPR = WN_OUT / 4;      // lanes per output row
    constexpr int RPI = 64 / LPR;        // rows per pass
    const int rr0 = lane / LPR, c4 = (lane % LPR) * 4;
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
        const int rr = it * RPI + rr0;
        const int gm = m0 + wr * 64 + rr, gc = ncol0 + c4;
        if (gm < M && gc < N) {
            f32x4 v = *(const f32x4*)(wst + rr * STG + c4);
            if (p.residual != nullptr) {
                const u32x2 rv = *(const u32x2*)(p.residual + (int64_t)(p.res_mod > 0 ? gm % p.res_mod : gm) * p.ldr + gc);
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

template <int EPI, bool OUT_F32, int WM, int NF>
static int launch_cfg(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = WM * 64, BNT = 32 * NF;
    constexpr int BN_OUT = (EPI == EPI_GATEUP) ? BNT / 2 : BNT;
    const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN_OUT);
    const size_t lds_main = (size_t)2 * (BM + BNT) * BK * 2, lds_epi = (size_t)WM * 2 * 64 * STG * sizeof(float);
    const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)gemm_bf16_tn<EPI, OUT_F32, WM, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_tn<EPI, OUT_F32, WM, NF>), dim3(tiles_m * tiles_n), dim3(WM * 128), lds, s, a, tiles_m);
    VILA_LAUNCH_CHECK();
    return 0;
}

bool gemm256_supported(const GemmArgs& a);
int gemm256_tiles_m_of(int M);        // 256-row tiles under the extra-row-fragment policy (gemm256_kernel.h: M = 256 k + r, r <= 16 -> k tiles)
int launch_gemm256(const GemmArgs& a, hipStream_t s);
int launch_gemm256_splitk(const GemmArgs& a, int splits, float* slab, hipStream_t s);
bool gemm_ring_supported(const GemmArgs& a);
int launch_gemm_ring(const GemmArgs& a, int stages, hipStream_t s);
int gemm_ring_splitk_slices(const GemmArgs& a);      // gemm_ring_splitk.hip: K-sliced 128x64 ring for short prompts (measured in round 5: the default for M < 512; VILA_RING_SPLITK=0 turns it off)
int launch_gemm_ring_splitk(const GemmArgs& a, int splits, hipStream_t s);
static int g_force_tile = 0;   // test / tuning hook: 0 auto, 1 = 128x128, 2 = 128x64, 3 = 256x128, 4 = 256x256 LDS-DMA kernel, 5 = split-K, 6 / 7 = 128x64 DMA ring with 4 / 3 stages, 8 = 128x128 DMA ring (2 stages), 11 = K-sliced 128x64 ring (needs a workspace), 12 / 13 / 14 = rings 7 / 8 / 6 with the PIPE 2 fragment schedule whatever VILA_RING_PIPE says, 15 / 16 / 17 = the same three with the plain schedule
extern "C" void vila_gemm_force_tile(int t) { g_force_tile = t; }
static int ring_splitk_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_RING_SPLITK"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}
// Measured and rejected (tools/gemm_bench pol, profiles/r02_gemm_bench_policies.log): slicing K four ways for ONE under-filled round with a
// long contraction (SFT down_proj forward / dgrad of gate and up: 182 tiles, 296 K-tiles -> 728 blocks = 2.84 rounds of a quarter of the
// work).  The slabs (4 x 44 MB written and read) eat the gain: 488 -> 481, 522 -> 545, 559 -> 544 us.
template <int EPI, bool OUT_F32>
static int launch_t(const GemmArgs& a, hipStream_t s) {
    int sel = g_force_tile;
    const int64_t tiles256 = (int64_t)gemm256_tiles_m_of(a.M) * cdiv(a.N, (EPI == EPI_GATEUP) ? 128 : 256);
    if (sel == 4 || (sel == 0 && tiles256 >= 150)) {
        if (gemm256_supported(a)) return launch_gemm256(a, s);
        if (sel == 4) sel = 0;
    }
    // under-filled grid of 256^2 tiles (S = 769 prefill: 56 tiles for N = 3584): slice K over grid.y when a workspace is given
    // (M >= 512 for the short contractions; a LONG contraction — down_proj, K = 18944 — is sliced at ANY M: at S = 289 (one image + a 32-token
    // prompt, BASELINE configs[1]'s short prompt) the ring kernel walked 296 K-tiles per block, 149 us per layer, where 2 x 14 tiles x 8 slices
    // take 57 + 11 us incl. the reduce: TTFT 13.2 -> 12.0 ms, profiles/r04_ttft_s289_ab.log; text-only prompts of 64 / 160 rows:
    // profiles/r04_gemm_bench_presmall.log)
    const int kt_all_ = cdiv(a.K, 64);
    if ((sel == 0 || sel == 5) && EPI == EPI_NONE && !OUT_F32 && a.ws != nullptr && (a.M >= 512 || kt_all_ >= 128) && gemm256_supported(a)) {
        const int kt = kt_all_;
        // as many K-slices as keep every block resident at once (one 512-thread block per CU), at most 8, at least 8 K-tiles each;
        // slices need not be equal (the last one takes the remainder): 42 tiles x 6 slices fills 252 CUs where 4 would fill 168
        int splits = (int)(256 / tiles256);
        if (splits > 8) splits = 8;
        while (splits >= 2 && (cdiv(kt, splits) < 8 || (size_t)splits * a.M * a.N * 4 > a.ws_bytes)) --splits;
        if (splits >= 2) splits = cdiv(kt, cdiv(kt, splits));      // drop empty trailing slices
        if (splits < 2) splits = 0;
        // K < 8192 (o_proj at S = 769): the DMA ring below does it in one launch at 557 TF/s vs 482 incl. the reduce
        // ... unless the slices are many and still long (ViT fc2 of one image, K = 4304: 20 tiles x 8 slices; with COLD weights — what a
        // forward pass sees — 45.6 -> 33.7 us, tools/gemm_bench precold; the warm numbers above hide that a lone tile's K loop runs at
        // HBM latency)
        // ... or the caller offers the next block's normalisation (o_proj of the prefill -> post-attention RMSNorm): the reduce then replaces
        // the norm launch as well (ring 44.4 + norm 7.7 us against slices + fused reduce, round 6)
        const bool norm_offer = a.norm_out != nullptr && a.norm_w != nullptr && a.N % 8 == 0 && a.N <= 16384 && splits >= 4;
        // ... or the q/k/v projection's RoPE + KV scatter (72 tiles -> 54 with the extra row fragment, 4 slices: the reduce replaces rope_kv_kernel)
        const bool rope_offer = gemm_rope_offer(a) && splits >= 4;
        if (kt < 128 && sel == 0 && !(kt >= 64 && splits >= 6) && !norm_offer && !rope_offer) splits = 0;
        // measured at M = 769 (tools/microbench.py prefill): N=3584,K=18944 233 -> 122 us; N=3584,K=3584 51 -> 41 us;
        // N=4608 (72 tiles) only breaks even, so require at least 4 slices
        if (splits >= 4 || (splits && sel == 5)) return launch_gemm256_splitk(a, splits, a.ws, s);
    }
    if (sel == 5) sel = 0;
    if constexpr (EPI == EPI_NONE && !OUT_F32) {
        if ((sel == 11 || (sel == 0 && ring_splitk_env() && a.M < 512)) && gemm_ring_supported(a)) {
            const int sp = gemm_ring_splitk_slices(a);
            if (sp >= 2) return launch_gemm_ring_splitk(a, sp, s);
        }
    }
    // everything below the gemm256 threshold: the LDS-DMA ring kernels (gemm_ring.hip) instead of the one-tile-ahead register staging
    if (EPI != EPI_GATEUP && !OUT_F32 && gemm_ring_supported(a)) {
        const int64_t tiles_ring = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 64);
        // measured (tools/microbench.py tiles): the 128x64 3-stage ring wins while its grid fits about one round of the 512
        // resident blocks (S=769 q/k/v 473 -> 675 TF/s, o_proj 397 -> 557, ViT fc2 168 -> 297); beyond that the 128x128 2-stage
        // ring takes over from the register-staged 128x128 kernel (SFT ViT shapes 376-590 -> 459-697, 4096^3 810 -> 1015)
        const int64_t tiles128r = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128);
        if (sel >= 12 && sel <= 14) return launch_gemm_ring(a, sel == 12 ? 203 : sel == 13 ? 208 : 204, s);   // PIPE = 2 explicitly
        if (sel >= 15 && sel <= 17) return launch_gemm_ring(a, sel == 15 ? 303 : sel == 16 ? 308 : 304, s);   // plain schedule explicitly
        if (sel == 7 || (sel == 0 && tiles_ring <= 560 && tiles128r < 270)) return launch_gemm_ring(a, 3, s);
        if (sel == 6) return launch_gemm_ring(a, 4, s);
        if (sel == 8 || sel == 0) return launch_gemm_ring(a, 8, s);
    }
    if (sel >= 6 && sel <= 17) sel = 0;
    if (sel == 0) {
        const int64_t tiles128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, (EPI == EPI_GATEUP) ? 64 : 128);
        if (tiles128 < 320 && EPI != EPI_GATEUP) sel = 2;
        else sel = 1;
    }
    if (sel == 2 && EPI == EPI_GATEUP) sel = 1;
    if (sel == 3) return launch_cfg<EPI, OUT_F32, 4, 4>(a, s);
    if constexpr (EPI != EPI_GATEUP) {
        if (sel == 2) return launch_cfg<EPI, OUT_F32, 2, 2>(a, s);
    }
    return launch_cfg<EPI, OUT_F32, 2, 4>(a, s);
}

// dgrad / wgrad on the tensors as they lie (contraction-major operands, gemm256_kernel.h): always the 256x256 kernel, sliced over K
// when its tiles cannot fill the chip and the caller lent a workspace
static int launch_gemm_cm(const GemmArgs& a, hipStream_t s) {
    VILA_REQUIRE(a.epi == EPI_NONE && !a.out_f32 && a.W2 == nullptr, "gemm: contraction-major operands take the plain bf16 epilogue only");
    VILA_REQUIRE(((uintptr_t)a.A % 16 == 0) && ((uintptr_t)a.W % 16 == 0) && ((uintptr_t)a.C % 16 == 0) && a.lda % 8 == 0 && a.ldw % 8 == 0 &&
                 a.ldc % 4 == 0 && a.N % 4 == 0, "gemm: pointers / leading dims must keep 16-B row alignment");
    VILA_REQUIRE(a.residual == nullptr || (a.ldr % 4 == 0 && (uintptr_t)a.residual % 8 == 0), "gemm: residual alignment");
    VILA_REQUIRE(gemm256_supported(a), "gemm: contraction-major operand needs rows %% 8 == 0, K >= 128 (M=%d N=%d K=%d a_cm=%d b_cm=%d)",
                 a.M, a.N, a.K, a.a_cm, a.b_cm);
    const int64_t tiles256 = (int64_t)cdiv(a.M, 256) * cdiv(a.N, 256);
    if (a.ws != nullptr && tiles256 < 150) {
        const int kt = cdiv(a.K, 64);
        int splits = (int)(256 / tiles256);
        if (splits > 8) splits = 8;
        while (splits >= 2 && (cdiv(kt, splits) < 8 || (size_t)splits * a.M * a.N * 4 > a.ws_bytes)) --splits;
        if (splits >= 2) splits = cdiv(kt, cdiv(kt, splits));
        if (splits >= 2) return launch_gemm256_splitk(a, splits, a.ws, s);
    }
    return launch_gemm256(a, s);
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
    VILA_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    if (a.a_cm || a.b_cm) return launch_gemm_cm(a, s);
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
