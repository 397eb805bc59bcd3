// 128x64x64 / 128x128x64 bf16 MFMA GEMM with an LDS-DMA ring, for the SHORT contractions of the path: the ViT blocks (M = 1024 per
// image, K = 1152 / 4304), the mm_projector and the S = 769 q/k/v projection.  
//
// Why a third kernel: those GEMMs are 3-10 GFLOP with 18-67 K-tiles.  gemm.hip stages HBM -> VGPR -> LDS with ONE tile of
// prefetch, so every K-tile pays a full memory round trip (~0.7 us measured per tile, 7 % of the MFMA rate), and gemm256's
// 256^2 tiles leave 90 % of the CUs idle (M = 1024, N = 1152 is 20 tiles).  Here the tile is small enough to spread over the
// chip (128x64: 144 blocks for that shape) and the K loop keeps THREE K-tiles in flight: `global_load_lds` (16 B per lane, no
// VGPR round trip) fills a ring of 4 stages x (A 16 KB + B 8 KB), counted `s_waitcnt vmcnt` releases tile t while t+1 and
// t+2 are still landing, one barrier per K-tile.
//   4 waves = 2 (M) x 2 (N), wave tile 64 x 32 = 4 x 2 accumulator fragments of v_mfma_f32_16x16x32_bf16, 16 MFMAs per wave
//   and K-tile against 12 ds_read_b128: LDS-read-bound at ~2/3 of the MFMA rate — fine for GEMMs this small, and the reason
//   the big contractions stay on gemm256 (0.375 reads per MFMA).
//   LDS image is lane-linear for the DMA; the 16-B slot swizzle (slot ^= (row >> 1) & 7, conflict-free for ds_read_b128, same
//   involution as gemm.hip / gemm256.hip) is applied on the per-lane SOURCE address and on the fragment reads.
//   K tail: chunks of the last K-tile beyond K are DMA'd from a zero chunk.  M / N tails: row clamping + masked stores.
// Epilogues: bias, GELU (tanh / erf), residual, bf16 out.
#include <stdlib.h>
#include "kernels.h"

#define RG_BM 128
#define RG_BK 64

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __attribute__((aligned(16))) unsigned int g_ring_zero_chunk[4];   // K-tail source (zero-initialised)

// PIPE = 2: fragment read the compiler's wait insertion does not see (retired by common.h lds_wait)
template <int OFF> __device__ __forceinline__ u32x4 ring_ds_read_b128(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

// NF = 16-column fragments per wave: NF = 2 -> 128 x 64 tile (24 KB per stage), NF = 4 -> 128 x 128 tile (32 KB per stage, wave
// tile 64 x 64: 16 ds_read_b128 per 32 MFMAs).  RG_STAGES x stage bytes <= 72 KB keeps two blocks per CU, so a second wave per
// SIMD overlaps one block's LDS reads with the other's MFMAs:  <NF=2, 3 stages> for grids of about one round (two K-tiles in
// flight per block), <NF=4, 2 stages> for the mid-size GEMMs of the SFT step's ViT (M = 4096, N, K in 1152..4304).
// PIPE = 2 (the default since round 5; PIPE = 0 = the plain schedule, VILA_RING_PIPE=0): left to itself the compiler issues the 12 fragment reads of
// a K-tile in FOUR groups, each followed by `s_waitcnt lgkmcnt(0)` and 4 MFMAs (it schedules for the smallest register footprint: 82 VGPRs where
// 256 are free at two waves per SIMD), so every K-tile exposes the LDS round trip four times.  With PIPE = 2 the fragment reads are inline-asm
// `ds_read_b128` (invisible to the compiler's wait insertion) issued ks-major and retired by register-tied `s_waitcnt lgkmcnt` (common.h lds_wait):
// the ks = 0 MFMAs start when the first 4 + NF reads are back and the ks = 1 reads land under them; the epilogue's bias / residual are requested
// up front instead of one dependent round trip per store pass.  Same arithmetic in the same order: results are bit-identical (tests/test_gpu_run.py).
// Measured (tools/gemm_bench prering, cold weights, profiles/r05_gemm_bench_prering.log): S = 769 qkv 44.8 -> 43.6 us, o_proj + residual 43.2 -> 42.4,
// ViT qkv 18.7 -> 18.3, ViT out_proj + residual 14.2 -> 12.8; TTFT 15.53 -> 15.33 ms (profiles/r05_second_call_ab.log).  Two things built with it at
// the end of round 4 (no GPU then) measured SLOWER or equal in the same log and are gone: a compiler-scheduled variant of the same order (PIPE = 1:
// 44.7 / 43.3 us) and 128x128 tiles with 3 / 4 stages (46.3 / 45.1 us for qkv, 48.3 / 47.9 for o_proj: one block per CU lacks waves, not stages).
template <int EPI, int RG_STAGES, int NF, int PIPE = 0>
__global__ __launch_bounds__(256, (RG_STAGES * (RG_BM + 32 * NF) * RG_BK * 2 <= 72 * 1024) ? 2 : 1) void gemm_ring_kernel(GemmArgs p, int tiles_m) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN = 32 * NF;
    constexpr int A_BYTES = RG_BM * RG_BK * 2;           // 16 KB
    constexpr int B_BYTES = BN * RG_BK * 2;              // 8 / 16 KB
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int B_IT = BN / 32;                        // DMA instructions per wave for the B tile (1 KB each)
    constexpr int DMA_PER_TILE = 4 + B_IT;
    constexpr int STG = 16 * NF + 4;                     // fp32 staging row stride (floats) of the epilogue
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = id % tiles_m, tn = id / tiles_m;
    const int m0 = tm * RG_BM, n0 = tn * BN;
    const int M = p.M, N = p.N, K = p.K;

    // ---- DMA source offsets: chunk c = (i * 4 + wave) * 64 + lane of a tile: row = c >> 3, LDS slot = c & 7 ----
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
    // row = c >> 3 = 8 * (i*4 + wave) + (lane >> 3)  =>  (row >> 1) & 7 = 4 * (wave & 1) + (lane >> 4), the same for every i
    const int kch_lane = (lane & 7) ^ (4 * (wave & 1) + (lane >> 4));
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);
    auto issue_tile = [&](int t) {
        const int k0 = t * RG_BK;
        char* base = smem + (t % RG_STAGES) * STAGE_BYTES + wave_lds;
        if (k0 + RG_BK <= K) {                              // block-uniform
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
                const bf16_t* src = kin ? p.A + aoff[i] + k0 : (const bf16_t*)g_ring_zero_chunk;
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(base + i * 4096), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const bf16_t* src = kin ? p.W + boff[i] + k0 : (const bf16_t*)g_ring_zero_chunk;
                __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(base + A_BYTES + i * 4096), 16, 0, 0);
            }
        }
    };

    // ---- fragment read offsets (bytes inside a tile) ----
    const int swr = (l15 >> 1) & 7;
    int foff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) foff[ks] = l15 * 128 + (((ks * 4 + lg) ^ swr) << 4);

    f32x4 acc[4][NF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = (K + RG_BK - 1) / RG_BK;
#pragma unroll
    for (int t = 0; t < RG_STAGES - 1; ++t)
        if (t < nt) issue_tile(t);
    for (int t = 0; t < nt; ++t) {
        // own DMAs of tile t have landed; up to RG_STAGES - 2 later tiles may still be in flight (DMA_PER_TILE per tile and lane)
        const int later = (nt - 1 - t) < (RG_STAGES - 2) ? (nt - 1 - t) : (RG_STAGES - 2);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DMA_PER_TILE) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                       // everyone's have; everyone has left tile t-1 (its stage is free)
        asm volatile("" ::: "memory");
        if (t + RG_STAGES - 1 < nt) issue_tile(t + RG_STAGES - 1);   // into the stage tile t-1 occupied
        const char* cA = smem + (t % RG_STAGES) * STAGE_BYTES + wr * 64 * 128;
        const char* cB = smem + (t % RG_STAGES) * STAGE_BYTES + A_BYTES + wc * (16 * NF) * 128;
        bf16x8 af[4][2], bfr[NF][2];
        if constexpr (PIPE == 2) {
            u32x4 ra[4][2], rb[NF][2];
            const uint32_t aA = lds_addr(cA), aB = lds_addr(cB);
            // issue order: the 4 + NF operands of ks = 0, then those of ks = 1.  lgkmcnt is a 4-bit counter: never more than 15 reads in flight, so
            // with NF = 4 (16 reads) the last ks = 1 read is issued behind the first wait.
            static_for<0, NF>([&](auto J) { constexpr int j = decltype(J)::value; rb[j][0] = ring_ds_read_b128<j * 2048>(aB + foff[0]); });
            static_for<0, 4>([&](auto I) { constexpr int i = decltype(I)::value; ra[i][0] = ring_ds_read_b128<i * 2048>(aA + foff[0]); });
            static_for<0, NF>([&](auto J) { constexpr int j = decltype(J)::value; rb[j][1] = ring_ds_read_b128<j * 2048>(aB + foff[1]); });
            static_for<0, 3>([&](auto I) { constexpr int i = decltype(I)::value; ra[i][1] = ring_ds_read_b128<i * 2048>(aA + foff[1]); });
            if constexpr (NF == 2) {
                ra[3][1] = ring_ds_read_b128<3 * 2048>(aA + foff[1]);                     // 12 in flight
                lds_wait<6>(rb[0][0], rb[1][0], ra[0][0], ra[1][0]); lds_wait<6>(ra[2][0], ra[3][0]);
            } else {
                lds_wait<7>(rb[0][0], rb[1][0], rb[2][0], rb[3][0], ra[0][0], ra[1][0], ra[2][0], ra[3][0]);   // 15 in flight, the first 8 back
                ra[3][1] = ring_ds_read_b128<3 * 2048>(aA + foff[1]);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra[i][0]), __builtin_bit_cast(bf16x8, rb[j][0]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);       // keep the ks = 0 MFMAs in front of the wait for ks = 1 (the scheduler otherwise sinks them behind it)
            if constexpr (NF == 2) { lds_wait<0>(rb[0][1], rb[1][1], ra[0][1], ra[1][1]); lds_wait<0>(ra[2][1], ra[3][1]); }
            else { lds_wait<0>(rb[0][1], rb[1][1], rb[2][1], rb[3][1], ra[0][1], ra[1][1], ra[2][1], ra[3][1]); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ra[i][1]), __builtin_bit_cast(bf16x8, rb[j][1]), acc[i][j], 0, 0, 0);
            continue;
        }
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bfr[j][ks] = *(const bf16x8*)(cB + j * 16 * 128 + foff[ks]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const bf16x8*)(cA + i * 16 * 128 + foff[ks]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][ks], bfr[j][ks], acc[i][j], 0, 0, 0);
    }
    __syncthreads();   // all LDS reads of the last tile done before the ring is reused as staging

    // ---- epilogue: two passes of 32 rows through per-wave fp32 staging [32][STG], then row-coalesced bf16 stores ----
    float* wst = (float*)smem + wave * 32 * STG;
    const int ncol0 = n0 + wc * (16 * NF);
    float bv[NF];
    if constexpr (!PIPE) {
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int col = ncol0 + j * 16 + l15;
        bv[j] = (p.bias != nullptr && col < N) ? bf2f(p.bias[col]) : 0.f;
    }
    }
    constexpr int LPR = 4 * NF, RPI = 64 / LPR;             // lanes per row (4 floats each), rows per store instruction
    const int rr0 = lane / LPR, c4 = (lane % LPR) * 4;
    constexpr int NIT = 32 / RPI;
    // PIPE: the epilogue's operands are requested up front.  Left as written, each of the 2 x NIT store passes below is
    // `global_load_dwordx2 (residual) -> s_waitcnt vmcnt(0) -> add -> global_store` and the bias is `global_load_ushort -> vmcnt(0)` per fragment:
    // 8-16 dependent round trips in the tail of every o_proj / out_proj / fc2 block (gfx950 counts stores in vmcnt too, so each wait also sits out the
    // previous pass's store).  A thread only ever reads the residual elements it writes itself, so an in-place residual stream stays correct.
    u32x2 rres[PIPE ? 2 * NIT : 1];
    if constexpr (PIPE) {
        bf16_t braw[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int col = ncol0 + j * 16 + l15;
            braw[j] = (p.bias != nullptr) ? p.bias[col < N ? col : 0] : (bf16_t)0;
        }
        if (p.residual != nullptr) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int gm = m0 + wr * 64 + h * 32 + it * RPI + rr0, gc = ncol0 + c4;
                    const bool ok = gm < M && gc < N;          // out-of-range slots fetch element (0, 0) and are never used
                    const int gmr = ok ? (p.res_mod > 0 ? gm % p.res_mod : gm) : 0;
                    rres[h * NIT + it] = *(const u32x2*)(p.residual + (int64_t)gmr * p.ldr + (ok ? gc : 0));
                }
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            uint32_t bb = braw[j];
            asm volatile("" : "+v"(bb));                       // (pins the conversion behind the requests above)
            bv[j] = (ncol0 + j * 16 + l15 < N) ? bf2f((bf16_t)bb) : 0.f;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[2 * h + ii][j][r] + bv[j];
                    if constexpr (EPI == EPI_GELU_TANH) v = gelu_tanh_f(v);
                    if constexpr (EPI == EPI_GELU_ERF) v = gelu_erf_f(v);
                    wst[(ii * 16 + lg * 4 + r) * STG + j * 16 + l15] = v;
                }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int rr = it * RPI + rr0;
            const int gm = m0 + wr * 64 + h * 32 + rr, gc = ncol0 + c4;
            if (gm < M && gc < N) {
                f32x4 v = *(const f32x4*)(wst + rr * STG + c4);
                if (p.residual != nullptr) {
                    u32x2 rv;
                    if constexpr (PIPE) rv = rres[h * NIT + it];
                    else rv = *(const u32x2*)(p.residual + (int64_t)(p.res_mod > 0 ? gm % p.res_mod : gm) * p.ldr + gc);
                    v[0] += lo_bf(rv[0]); v[1] += hi_bf(rv[0]); v[2] += lo_bf(rv[1]); v[3] += hi_bf(rv[1]);
                }
                u32x2 o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
                *(u32x2*)((bf16_t*)p.C + (int64_t)gm * p.ldc + gc) = o;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int EPI, int STAGES, int NF, int PIPE = 0>
static int launch_ring_t(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = cdiv(a.M, RG_BM), tiles_n = cdiv(a.N, 32 * NF);
    const size_t lds = (size_t)STAGES * (RG_BM + 32 * NF) * RG_BK * 2;    // >= 4 waves x 32 x (16 NF + 4) x 4 B of staging
    static bool attr_set = false;
    if (!attr_set) {
        VILA_HIP(hipFuncSetAttribute((const void*)gemm_ring_kernel<EPI, STAGES, NF, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<EPI, STAGES, NF, PIPE>), dim3(tiles_m * tiles_n), dim3(256), lds, s, a, tiles_m);
    VILA_LAUNCH_CHECK();
    return 0;
}

bool gemm_ring_supported(const GemmArgs& a) {
    return a.epi != EPI_GATEUP && !a.out_f32 && a.K % 8 == 0 && a.K >= RG_BK && a.N % 4 == 0 && (int64_t)a.M * a.lda < (1ll << 31) &&
           (int64_t)a.N * a.ldw < (1ll << 31);
}

template <int STAGES, int NF, int PIPE = 0>
static int launch_ring_epi(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_NONE: return launch_ring_t<EPI_NONE, STAGES, NF, PIPE>(a, s);
        case EPI_GELU_TANH: return launch_ring_t<EPI_GELU_TANH, STAGES, NF, PIPE>(a, s);
        case EPI_GELU_ERF: return launch_ring_t<EPI_GELU_ERF, STAGES, NF, PIPE>(a, s);
    }
    VILA_FAIL(-1, "gemm_ring: unsupported epilogue %d", a.epi);
}

// variant: 3 = 128x64 tile, 3 stages (2 blocks / CU); 4 = 128x64, 4 stages (1 block / CU); 8 = 128x128 tile, 2 stages (2 blocks / CU).
// + 200 = PIPE 2 explicitly, + 300 = the plain fragment schedule explicitly (tests / tools/gemm_bench); otherwise VILA_RING_PIPE decides: "0" = the plain schedule, anything else (or unset) = PIPE 2 — the compiler-scheduled PIPE 1 of round 4 was
// measured and deleted in round 5, so VILA_RING_PIPE=1 means PIPE 2 today.
static int ring_pipe_env() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_RING_PIPE"); v = (e && e[0] == '0') ? 0 : 2; }
    return v;
}
int launch_gemm_ring(const GemmArgs& a, int variant, hipStream_t s) {
    const int pipe = variant >= 300 ? 0 : variant >= 200 ? 2 : ring_pipe_env();
    variant %= 100;
    if (pipe) {
        if (variant == 8) return launch_ring_epi<2, 4, 2>(a, s);
        if (variant == 4) return launch_ring_epi<4, 2, 2>(a, s);
        return launch_ring_epi<3, 2, 2>(a, s);
    }
    if (variant == 8) return launch_ring_epi<2, 4>(a, s);
    if (variant == 4) return launch_ring_epi<4, 2>(a, s);
    return launch_ring_epi<3, 2>(a, s);
}
