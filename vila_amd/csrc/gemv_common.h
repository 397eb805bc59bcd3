// Shared pieces of the decode GEMV kernels (gemv.hip: bf16 weights, gemv_w4.hip: int4 group-quantised weights).
#pragma once
#include "kernels.h"

// ---- chained kernels (kernels.h ChainLink) -----------------------------------------------------------------------------------------------
// chain_wait: thread 0 polls the predecessor's done counter (bounded: ~0.3 s), the block meets at a barrier, every wave then takes an
// agent-scope acquire (the predecessor ran on other CUs / XCDs: its stores were written back by chain_done's release).
__device__ __forceinline__ void chain_wait(const ChainLink& c) {
    if (c.ctr == nullptr || c.wait_idx < 0) return;
    if (threadIdx.x == 0) {
        // once a wait has given up, the token is lost anyway: later waits return at once, so a scheduling surprise costs ~0.2 s, not a hang
        if (__hip_atomic_load(c.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            uint32_t n = 0;
            while (__hip_atomic_load(c.ctr + c.wait_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c.wait_target) {
                __builtin_amdgcn_s_sleep(4);
                if (++n > (1u << 18)) { __hip_atomic_store(c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// chain_done: after the block's last store.  The barrier retires every wave's stores to L2 (workgroup release), thread 0 then releases at
// agent scope (L2 write-back) and counts the block.
__device__ __forceinline__ void chain_done(const ChainLink& c) {
    if (c.ctr == nullptr || c.done_idx < 0) return;
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c.ctr + c.done_idx, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- activation staging -------------------------------------------------------------------------
// stage x (optionally RMS-normalised with gain, HF rounding order) as bf16 into LDS; all 256 threads participate.
// Single pass for K <= 8192 (x kept in registers between the sum of squares and the scaling).
__device__ __forceinline__ void stage_x(const bf16_t* __restrict__ x, const bf16_t* __restrict__ norm_w, float eps, int K,
                                        bf16_t* sx, float* scratch) {
    const int tid = threadIdx.x, nch = K >> 3;
    if (norm_w == nullptr) {
        // 4 independent 16-B loads in flight per thread and pass (K = 18944: 3 passes instead of 10 dependent round trips)
        for (int c0 = tid; c0 < nch; c0 += 1024) {
            u32x4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int c = c0 + 256 * i; t[i] = (c < nch) ? *(const u32x4*)(x + c * 8) : (u32x4){0u, 0u, 0u, 0u}; }
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
            v[i] = (c < nch) ? *(const u32x4*)(x + c * 8) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float a = lo_bf(v[i][k]), b = hi_bf(v[i][k]); s += a * a + b * b; }
        }
    } else {
        for (int c = tid; c < nch; c += 256) {
            const u32x4 t = *(const u32x4*)(x + c * 8);
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
            const u32x4 t = *(const u32x4*)(x + c * 8);
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

