// Shared pieces of the decode GEMV kernels (gemv.hip: bf16 weights, gemv_w4.hip: int4 group-quantised weights).
#pragma once
#include "kernels.h"

// ---- chained kernels (kernels.h ChainLink) -----------------------------------------------------------------------------------------------
// The hand-off is FENCE-FREE (guide G16 "R1": payload stored write-through, consumer loads sc1): an agent-scope release writes back the XCD's
// whole L2 and an acquire per block costs ~1.7 us (x4 at 4 blocks per CU) — the first version of this chain, with both, ran the token 2.7x
// SLOWER than the plain step, and 512 blocks polling every 0.1 us took most of the predecessor's HBM bandwidth on top.  Now
//   producer : outputs as relaxed agent-scope atomic stores (global_store ... sc1: write-through, the line leaves the L2) -> every wave drains
//              its stores (s_waitcnt vmcnt(0)) -> barrier -> ONE lane counts the block with a relaxed agent-scope atomic add
//   consumer : ONE lane sleeps through the predecessor's predicted run time, then polls the counter (relaxed, s_sleep between polls) ->
//              barrier -> the activation is read with sc1 loads (L1 bypassed; the producer's sc1 stores dropped the L2 copies)
typedef __attribute__((address_space(1))) uint32_t chain_gu32;
typedef __attribute__((address_space(1))) unsigned short chain_gu16;
// A poll is a fabric read of ONE word: 512 blocks polling the same counter made its memory channel a hot spot every wave of the co-running
// predecessor had to queue behind (third version of this chain: main phases 1.7-3.6x slower with a sleeping successor resident).  So the
// arrival COUNT is only ever added to (one atomic per block), the block that completes it raises CHAIN_FLAGS replicated flag words 256 B apart,
// and block b polls flag b % CHAIN_FLAGS: 8 pollers per word and channel.
__device__ __forceinline__ bool chain_poll(const ChainLink& c) {
    const uint32_t* f = c.ctr + (size_t)c.wait_idx * CHAIN_WORDS + (1 + (blockIdx.x % CHAIN_FLAGS)) * CHAIN_STRIDE;
    return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
// The wait is mostly SLEEP: the host passes the predecessor's predicted run time; polls come at 1/2, 3/4, 7/8 ... of it (a kernel that
// finishes early is noticed within 1/8 of its run time, ~6 polls per block before the prediction), then every ~0.35 us.
__device__ __forceinline__ void chain_wait(const ChainLink& c) {
    if (c.ctr == nullptr || c.wait_idx < 0) return;
    if (threadIdx.x == 0) {
        bool done = false;
        int rem = c.pre_sleep_us;
        while (rem > 1 && !done) {
            const int nap = rem >> 1;
            for (int i = 0; i < nap; ++i) __builtin_amdgcn_s_sleep(36);                  // ~1 us each
            rem -= nap;
            done = chain_poll(c);
        }
        uint32_t n = 0;
        while (!done) {
            done = chain_poll(c);
            if (done) break;
            __builtin_amdgcn_s_sleep(12);                                                // ~0.35 us between polls
            ++n;
            // once a wait has given up the token is lost anyway: later waits return at once, so a scheduling surprise costs ~0.2 s, not a hang
            if ((n & 127u) == 0u && __hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            if (n > (1u << 19)) { __hip_atomic_store(c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}
// after the block's last (sc1) store
__device__ __forceinline__ void chain_done(const ChainLink& c) {
    if (c.ctr == nullptr || c.done_idx < 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     // every storing wave drains its write-through stores
    __syncthreads();
    uint32_t* base = c.ctr + (size_t)c.done_idx * CHAIN_WORDS;
    // wave 0: lane 0 counts the block; the block that completes the count raises the flags, one lane per flag
    if (threadIdx.x < 64) {
        uint32_t old = 0;
        if (threadIdx.x == 0) old = __hip_atomic_fetch_add(base, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __shfl(old, 0, 64);
        if (old + 1u == c.done_blocks && threadIdx.x < CHAIN_FLAGS)
            __hip_atomic_store(base + (1 + threadIdx.x) * CHAIN_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// two adjacent bf16 outputs (n even) as ONE store; coherent = write-through for a chained successor
__device__ __forceinline__ void store_bf16_pair(bf16_t* y, int n, bool two, bf16_t o0, bf16_t o1, bool coherent) {
    if (two) {
        const uint32_t w = (uint32_t)o0 | ((uint32_t)o1 << 16);
        if (coherent) __hip_atomic_store((uint32_t*)(y + n), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *(uint32_t*)(y + n) = w;
    } else {
        if (coherent) __hip_atomic_store((unsigned short*)(y + n), (unsigned short)o0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else y[n] = o0;
    }
}
// 16-byte load of chunk c of x: COH = through a buffer descriptor with the sc1 bit (aux 16): never served by this CU's L1
template <bool COH>
__device__ __forceinline__ u32x4 ldx16(const bf16_t* x, __amdgpu_buffer_rsrc_t rs, int c) {
    if constexpr (COH) return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)c * 16u, 0, 16));
    else return *(const u32x4*)(x + c * 8);
}

// ---- CU-balanced row-group map (round 6) -------------------------------------------------------------------------------------------
// The grid-stride walk of rounds 1-5 (group = block * 4 + wave, += grid * 4) balances WAVES, not CUs: blocks b and b + 256 share a CU under
// round-robin dispatch, so with 448 working blocks of a 512-block grid 192 CUs streamed 8 row pairs of down_proj and 64 CUs 4, and 64 CUs took
// 40 of gate/up's groups against 36 on the others — the kernel ends when the fullest CU does.  Here a CU owns groups {j * ncu + cu}: every
// shape of NVILA-8B divides evenly (37 / 7 / 9 / 297 groups per CU), and the waves of the CU's blocks deal its groups among themselves.
// On top, `skew` groups per CU move from the odd XCDs' CUs to the even ones' (the round-6 trace: blocks with an odd (block % 8) stream ~7 %
// slower): every CU takes cf = n / ncu - skew groups by the interleaved map, the rest goes to the even CUs only.
struct CuMap {
    int ncu, cf, n_fast, rem;
    __device__ __forceinline__ CuMap(int n_groups, int ncu_, int skew) {
        ncu = ncu_; n_fast = ncu_ >> 1;
        const bool can = (ncu_ & 7) == 0;
        cf = n_groups / ncu_ - (can ? skew : 0); cf = cf < 0 ? 0 : cf;
        rem = n_groups - cf * ncu_;
        if (!can) { cf = 0x3fffffff; rem = 0; }          // odd grids: the plain interleaved map (count() below handles the bound)
        n_total = n_groups;
    }
    int n_total;
    __device__ __forceinline__ int rank(int cu) const { return (cu >> 3) * 4 + ((cu & 7) >> 1); }
    __device__ __forceinline__ int count(int cu) const {
        if (cf == 0x3fffffff) return n_total > cu ? (n_total - cu + ncu - 1) / ncu : 0;
        const bool fast = (cu & 1) == 0;
        const int r = rank(cu);
        return cf + ((fast && rem > r) ? (rem - r + n_fast - 1) / n_fast : 0);
    }
    __device__ __forceinline__ int gid(int cu, int j) const {
        if (cf == 0x3fffffff || j < cf) return j * ncu + cu;
        return cf * ncu + (j - cf) * n_fast + rank(cu);
    }
};

// ---- activation staging -------------------------------------------------------------------------
// stage x (optionally RMS-normalised with gain, HF rounding order) as bf16 into LDS; all 256 threads participate.
// Single pass for K <= 8192 (x kept in registers between the sum of squares and the scaling).
template <bool COH = false>
__device__ __forceinline__ void stage_x(const bf16_t* x, const bf16_t* __restrict__ norm_w, float eps, int K,
                                        bf16_t* sx, float* scratch) {
    const int tid = threadIdx.x, nch = K >> 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, COH ? K * 2 : 0, 0x00020000);
    if (norm_w == nullptr) {
        // 4 independent 16-B loads in flight per thread and pass (K = 18944: 3 passes instead of 10 dependent round trips)
        for (int c0 = tid; c0 < nch; c0 += 1024) {
            u32x4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int c = c0 + 256 * i; t[i] = (c < nch) ? ldx16<COH>(x, rs, c) : (u32x4){0u, 0u, 0u, 0u}; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int c = c0 + 256 * i; if (c < nch) *(u32x4*)(sx + c * 8) = t[i]; }
        }
        __syncthreads();
        return;
    }
    constexpr int MAXC = 4;
    const bool small = nch <= 256 * MAXC;
    u32x4 v[MAXC];
    float s = 0.f;
    if (small) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = tid + 256 * i;
            v[i] = (c < nch) ? ldx16<COH>(x, rs, c) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float a = lo_bf(v[i][k]), b = hi_bf(v[i][k]); s += a * a + b * b; }
        }
    } else {
        for (int c = tid; c < nch; c += 256) {
            const u32x4 t = ldx16<COH>(x, rs, c);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float a = lo_bf(t[k]), b = hi_bf(t[k]); s += a * a + b * b; }
        }
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) scratch[tid >> 6] = s;
    __syncthreads();
    const float rstd = rsqrtf((scratch[0] + scratch[1] + scratch[2] + scratch[3]) / K + eps);
    if (small) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = tid + 256 * i;
            if (c < nch) {
                const u32x4 g = *(const u32x4*)(norm_w + c * 8);
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = pack2bf(lo_bf(g[k]) * bfround(lo_bf(v[i][k]) * rstd), hi_bf(g[k]) * bfround(hi_bf(v[i][k]) * rstd));
                *(u32x4*)(sx + c * 8) = o;
            }
        }
    } else {
        for (int c = tid; c < nch; c += 256) {
            const u32x4 t = ldx16<COH>(x, rs, c);
            const u32x4 g = *(const u32x4*)(norm_w + c * 8);
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = pack2bf(lo_bf(g[k]) * bfround(lo_bf(t[k]) * rstd), hi_bf(g[k]) * bfround(hi_bf(t[k]) * rstd));
            *(u32x4*)(sx + c * 8) = o;
        }
    }
    __syncthreads();
}

