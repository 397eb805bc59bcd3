// Batch-1 decode token as ONE persistent launch (round 6; SURVEY.md §8 row a12, reference call site llava_arch.py:823-833 ->
// GenerationMixin's one-token forward).  The 145-launch step (gemv.hip, api.hip decode_step_impl) is "bytes / 7.4 TB/s + ~8 us" per
// kernel: five all-to-all edges per layer, each paid as a kernel boundary whose HBM pipe runs dry (tail of the predecessor, dispatch,
// first round trip of the successor).  Earlier rounds priced the alternatives ONE AT A TIME and each lost: a grid barrier (5-6 us) costs
// more than the boundary (round 2), flag hand-offs between co-resident kernels cost three fabric round trips (round 4).  What none of
// them did is keep the WEIGHT STREAM running across the edge.  This kernel does:
//   * one block per CU, 7 worker waves + 1 sync wave, resident for the whole token (28 layers x 5 phases + lm_head);
//   * a phase's weights do not depend on activations, so every worker requests the first batch of its NEXT phase (14-28 x 16 B per
//     lane = 14-28 KB per wave, ~100-200 KB per CU, 25-50 MB chip-wide) before it reports the current phase done: the grid barrier's
//     latency (~4-5 us) is spent streaming, and the short phases (q/k/v 33 MB, o_proj 26 MB) have ALL their bytes in flight by the
//     time the barrier opens;
//   * the barrier itself is fence-free (guide G16 "R1"): a phase's outputs leave through the sync wave as write-through (sc1) stores,
//     which it drains before it counts the block; consumers read activations with sc1 loads.  The sync wave holds no prefetched loads,
//     so its polls are not queued behind them (loads return in order per wave) — that is why the roles are split;
//   * arithmetic is the launch path's, operation for operation (same per-lane chunk order, same wave reduction, same RMSNorm partial
//     sums, same 16-key chunk partials and merges in the attention), so the logits are BIT-IDENTICAL to the 145-launch step
//     (tests/test_gpu_model.py::test_persistent_decode_step_equals_the_launch_path).
// Every spin is bounded: a wait that gives up sets workspace word 0 (vila_llm_decode_chain_error) and the kernel runs to its end.
#include "kernels.h"
#include "gemv_common.h"
#include "decode_persist.h"
#include <stdlib.h>

#define DPW DP_WORKERS
typedef __attribute__((address_space(1))) unsigned long long dp_gu64;

// ---- block barrier: LDS traffic retired, no vmcnt wait (prefetched weight loads stay in flight across it) ----
#define DP_BB() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

__device__ __forceinline__ uint32_t dp_ld_u32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float dp_ld_f32(const float* p) { return __uint_as_float(__hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
// sc1 loads as BUFFER loads (aux 16): an atomic load is waited for on the spot (vmcnt(0) behind it — every prefetched weight load too), a
// buffer load is an ordinary load the compiler waits for at its first use
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dp_rsrc(const void* p, int bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000); }
__device__ __forceinline__ float dp_ld_bf16(__amdgpu_buffer_rsrc_t rs, int idx) {
    return bf2f((bf16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, (unsigned)idx * 2u, 0, 16));
}
__device__ __forceinline__ void dp_st_u32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dp_st_u16(bf16_t* p, bf16_t v) { __hip_atomic_store((unsigned short*)p, (unsigned short)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void dp_st_f2(float* p, float a, float b) {
    __hip_atomic_store((dp_gu64*)p, (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// weight loads: explicitly GLOBAL address space — through the row-pointer arrays the compiler loses the kernel-argument provenance and emits
// flat_load, which counts on vmcnt AND lgkmcnt and turns every counted wait of the ring into a full drain
typedef const __attribute__((address_space(1))) u32x4* dp_gptr;
__device__ __forceinline__ u32x4 dp_ldw(const bf16_t* p) { return __builtin_nontemporal_load((dp_gptr)(uintptr_t)p); }
__device__ __forceinline__ float dp_dot8(const u32x4 w, const u32x4 x, float acc) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        acc = fmaf(lo_bf(w[k]), lo_bf(x[k]), acc);
        acc = fmaf(hi_bf(w[k]), hi_bf(x[k]), acc);
    }
    return acc;
}

// ---- grid barrier, executed by the sync wave only ------------------------------------------------------------------------------
// Words (DP_SYNC_STRIDE apart, zeroed per token by the prologue kernel): arrival count of group g = ctr[g], top count = ctr[8],
// generation of group g = ctr[9 + g].  Counts only grow (barrier e completes a group at e * group_size, the top at e * n_groups), so
// nothing is ever reset and no ordering between a reset and a later arrival is needed.  Group = block % 8 (an XCD when blocks are
// dealt round-robin — a speed assumption only).
struct DpSync {
    uint32_t* err; uint32_t* ctr;
    uint32_t epoch, n_blocks, n_grp, grp;
    bool dead;
};
// Split barrier.  ARRIVE: every block adds 1 to each of the n_grp replicated counters (ONE wave instruction: lane g adds to counter g; the adds
// return nothing, so nothing is waited for).  WAIT: lane 0 polls the block's own replica (32 pollers per word) until it reads epoch * n_blocks.
// No last-arriver role, no flag stores, no returned atomic: the chain "last block's stores drained -> its adds land -> a poll sees them" is the
// whole latency, and between ARRIVE and WAIT the block's workers may start streaming the next phase's weights.
__device__ __forceinline__ void dp_arrive(DpSync& s, int lane) {
    s.epoch += 1u;
    if (s.dead) return;
    if (lane < (int)s.n_grp) (void)__hip_atomic_fetch_add(s.ctr + (size_t)lane * DP_SYNC_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dp_wait(DpSync& s, int lane) {
    if (s.dead) return;
    int dead = 0;
    if (lane == 0) {
        const uint32_t* cnt = s.ctr + (size_t)s.grp * DP_SYNC_STRIDE;
        const uint32_t target = s.epoch * s.n_blocks;
        uint32_t n = 0;
        while (dp_ld_u32(cnt) < target) {
            __builtin_amdgcn_s_sleep(1);
            ++n;
            if ((n & 255u) == 0u && dp_ld_u32(s.err) != 0u) { dead = 1; break; }
            if (n > (1u << 21)) { dp_st_u32(s.err, 1u); dead = 1; break; }
        }
    }
    dead = __shfl(dead, 0, 64);
    if (dead) s.dead = true;
}

// ---- weight batches ----------------------------------------------------------------------------------------------------------
// slot s = u * R + r holds chunk (c0 + 64 u + lane) of row r: the launch path's order (gemv.hip fma_batch: for u, for r), so every
// row's per-lane sum runs over its chunks in the same ascending order
// ONE buffer for every phase (a phase uses its first R * U slots): separate buffers per phase are loop-carried and conditionally written, so the
// register allocator has to keep all of them alive through the whole layer loop (56 + 112 + 64 VGPRs: spills)
#define DP_SLOTS 16
struct DpBufAll { u32x4 v[DP_SLOTS]; };
template <int R, int U> using DpBuf = DpBufAll;
// every slot (re)defined by an empty asm: ends the live range of whatever the slots held (the prefetches below are conditional, so without
// this the allocator must assume the old contents of all 28 slots are still wanted on the not-taken path — through the attention phase too)
__device__ __forceinline__ void dp_kill(DpBufAll& b) {
#pragma unroll
    for (int s = 0; s < DP_SLOTS; ++s) asm volatile("" : "=v"(b.v[s]));
}

template <int R, int U>
__device__ __forceinline__ void dp_load_batch(DpBuf<R, U>& b, const bf16_t* const (&rows)[R], int c0, int lane, int nch) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        // chunks beyond the row are CLAMPED, not predicated (their x chunk is zero in dp_consume): a predicated load is a branch per slot, and
        // behind a branch the compiler's wait counts collapse to vmcnt(0) — the whole prefetch ring would drain at every slot
        int c = c0 + u * 64 + lane; c = c < nch ? c : nch - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) b.v[u * R + r] = dp_ldw(rows[r] + (size_t)c * 8);
    }
}
// consume the batch at c0 and (REFILL) request the batch at (nrows, nc0) slot by slot behind it
template <int R, int U, bool REFILL>
__device__ __forceinline__ void dp_consume(DpBuf<R, U>& b, const bf16_t* sx, int c0, int lane, int nch, float (&acc)[R],
                                           const bf16_t* const (&nrows)[R], int nc0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = c0 + u * 64 + lane;
        u32x4 xv = *(const u32x4*)(sx + (c < nch ? c : nch - 1) * 8);
        if (c >= nch) xv = (u32x4){0u, 0u, 0u, 0u};
        int nc = nc0 + u * 64 + lane; nc = nc < nch ? nc : nch - 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = dp_dot8(b.v[u * R + r], xv, acc[r]);
            if constexpr (REFILL) {
                // The slot is refilled IN PLACE, behind its own use: the empty asm makes the refill's address depend on the sum just formed, so the
                // scheduler cannot hoist the load above the use and keep old and new slot alive together (which doubled the buffer: 2 x 112
                // VGPRs for the gate/up quads = spills)
                asm volatile("" : "+v"(nc) : "v"(acc[r]));
                b.v[u * R + r] = dp_ldw(nrows[r] + (size_t)nc * 8);
            }
        }
    }
}

// one GEMV phase of a worker wave: its groups li = w, w + DPW, ... < ng (local index: group id = li * NB + block);
// precondition: b holds batch 0 of group w (dp_load_batch issued earlier, possibly phases ago)
template <int R, int U, class RowsOf, class EpiFetch, class Finish>
__device__ __forceinline__ void dp_gemv_phase(DpBuf<R, U>& b, const bf16_t* sx, int nch, int ng, int w, int lane,
                                              RowsOf rows_of, EpiFetch epi_fetch, Finish finish) {
    if (w >= ng) return;
    const int nb = (nch + 64 * U - 1) / (64 * U);
    const bf16_t* rows[R];
    const bf16_t* nrows[R];
    int li = w;
    rows_of(li, rows);
    for (;;) {
        epi_fetch(li);
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        const int nli = li + DPW;
        const bool more = nli < ng;
        if (more) rows_of(nli, nrows);
        for (int j = 0; j + 1 < nb; ++j) dp_consume<R, U, true>(b, sx, j * 64 * U, lane, nch, acc, rows, (j + 1) * 64 * U);
        if (more) dp_consume<R, U, true>(b, sx, (nb - 1) * 64 * U, lane, nch, acc, nrows, 0);
        else dp_consume<R, U, false>(b, sx, (nb - 1) * 64 * U, lane, nch, acc, rows, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
        finish(li, acc);
        if (!more) break;
        li = nli;
#pragma unroll
        for (int r = 0; r < R; ++r) rows[r] = nrows[r];
    }
}

// ---- activation staging (all 8 waves call; workers act) -----------------------------------------------------------------------
// RMSNorm form: the launch path's stage_x<true> with its 256 threads = worker waves 0..3 (same chunk -> thread map, same partial
// sums, same order of the four wave sums): bit-identical rstd.
// RMSNorm form: the launch path's stage_x<true> with its 256 threads = worker waves 0..3 (same chunk -> thread map, same partial sums, same
// order of the four wave sums): bit-identical rstd.  The gain is requested together with x (the launch path reads it behind the reduction: one
// more dependent round trip).  (Requesting it ahead of the grid barrier, with the weight prefetch, was tried: the allocator spills it across
// the barrier and reloads it element by element.)
__device__ __forceinline__ void dp_stage_x_norm(const bf16_t* x, const bf16_t* __restrict__ norm_w, float eps, int K, bf16_t* sx, float* scratch, int tid) {
    const int nch = K >> 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, K * 2, 0x00020000);
    constexpr int MAXC = 4;                                      // K <= 8192 (launcher check)
    u32x4 v[MAXC], g[MAXC];
    float s = 0.f;
    if (tid < 256) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = tid + 256 * i;
            v[i] = (c < nch) ? ldx16<true>(x, rs, c) : (u32x4){0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < MAXC; ++i) { int c = tid + 256 * i; c = c < nch ? c : nch - 1; g[i] = *(const u32x4*)(norm_w + c * 8); }
#pragma unroll
        for (int i = 0; i < MAXC; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float a = lo_bf(v[i][k]), b = hi_bf(v[i][k]); s += a * a + b * b; }
        s = wave_sum(s);
        if ((tid & 63) == 0) scratch[tid >> 6] = s;
    }
    DP_BB();
    const float rstd = rsqrtf((scratch[0] + scratch[1] + scratch[2] + scratch[3]) / K + eps);
    if (tid < 256) {
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = tid + 256 * i;
            if (c < nch) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = pack2bf(lo_bf(g[i][k]) * bfround(lo_bf(v[i][k]) * rstd), hi_bf(g[i][k]) * bfround(hi_bf(v[i][k]) * rstd));
                *(u32x4*)(sx + c * 8) = o;
            }
        }
    }
    DP_BB();
}
__device__ __forceinline__ void dp_stage_x_copy(const bf16_t* x, int K, bf16_t* sx, int tid) {
    const int nch = K >> 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, K * 2, 0x00020000);
    if (tid < DPW * 64) {
        for (int c0 = tid; c0 < nch; c0 += 4 * DPW * 64) {
            u32x4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int c = c0 + DPW * 64 * i; t[i] = (c < nch) ? ldx16<true>(x, rs, c) : (u32x4){0u, 0u, 0u, 0u}; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int c = c0 + DPW * 64 * i; if (c < nch) *(u32x4*)(sx + c * 8) = t[i]; }
        }
    }
    DP_BB();
}
// x = merge of the attention partials over the active 256-key slices (gemv.hip stage_x_attn, same arithmetic per element); the partials
// of a head are requested together (sc1 buffer loads), not one dependent round trip per slice
#define DP_MAXS 8                                                // slices of 256 keys: caches up to 2048 positions
__device__ __forceinline__ void dp_stage_x_attn(const float* part_o, const float* part_ml, int n_active, int nq, bf16_t* sx, float* wsm, int tid) {
    if (tid < DPW * 64) {
        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)part_ml, 0, n_active * nq * 8, 0x00020000);
        for (int h = tid; h < nq; h += DPW * 64) {
            u32x2 ml[DP_MAXS];
#pragma unroll
            for (int s = 0; s < DP_MAXS; ++s) {
                const int sc = s < n_active ? s : n_active - 1;
                ml[s] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rm, (unsigned)((sc * nq + h) * 8), 0, 16));
            }
            float M = -INFINITY;
#pragma unroll
            for (int s = 0; s < DP_MAXS; ++s) if (s < n_active) M = fmaxf(M, __uint_as_float(ml[s][0]));
            float L = 0.f;
#pragma unroll
            for (int s = 0; s < DP_MAXS; ++s) if (s < n_active) L += __expf(__uint_as_float(ml[s][0]) - M) * __uint_as_float(ml[s][1]);
            const float invL = 1.f / L;
#pragma unroll
            for (int s = 0; s < DP_MAXS; ++s) if (s < n_active) wsm[s * nq + h] = __expf(__uint_as_float(ml[s][0]) - M) * invL;
        }
    }
    DP_BB();
    if (tid < DPW * 64) {
        const int n4 = nq * 32;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)part_o, 0, n_active * nq * 128 * 4, 0x00020000);
        for (int i = tid; i < n4; i += DPW * 64) {
            const int h = i >> 5;
            f32x4 pv[DP_MAXS];
#pragma unroll
            for (int s = 0; s < DP_MAXS; ++s) {
                const int sc = s < n_active ? s : n_active - 1;
                pv[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((sc * nq * 128 + i * 4) * 4), 0, 16));
            }
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < DP_MAXS; ++s) {
                if (s < n_active) {
                    const float wv = wsm[s * nq + h];
                    o[0] = fmaf(wv, pv[s][0], o[0]); o[1] = fmaf(wv, pv[s][1], o[1]); o[2] = fmaf(wv, pv[s][2], o[2]); o[3] = fmaf(wv, pv[s][3], o[3]);
                }
            }
            u32x2 r; r[0] = pack2bf(o[0], o[1]); r[1] = pack2bf(o[2], o[3]);
            *(u32x2*)(sx + i * 4) = r;
        }
    }
    DP_BB();
}

// ---- the kernel --------------------------------------------------------------------------------------------------------------
// outq entry = {index, payload}: what the sync wave stores after the block's workers are done with a phase
enum { DPK_QKV = 0, DPK_PAIR = 1, DPK_F32 = 2 };

__global__ __launch_bounds__(DP_THREADS, 2) void decode_token_kernel(DpArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_sync = wave == DPW;
    const int NB = gridDim.x, blk = blockIdx.x;
    const int H = p.H, F = p.F, hd = p.hd, half = hd >> 1;
    const int QS = p.nq * hd;
    bf16_t* sx = (bf16_t*)smem;
    float* scratch = (float*)(smem + p.lds_scratch);
    uint32_t* outq = (uint32_t*)(smem + p.lds_outq);
    float* sq = (float*)(smem + p.lds_attn);
    float* so = sq + 128;                 // [16][128]
    float* sml = so + 16 * 128;           // [16][2]
    float* fin = sml + 32;                // [128] o, [2] m, l
    float* wsm = (float*)(smem + p.lds_wsm);
    const int pos = *p.pos_ptr;
    const int nkeys_all = pos + 1;
    const int n_active = (pos + 256) / 256;

    DpSync sy;
    sy.err = p.sync; sy.ctr = p.sync + 64;
    sy.epoch = 0u; sy.n_grp = NB < 8 ? NB : 8; sy.grp = blk % sy.n_grp; sy.n_blocks = NB; sy.dead = false;

    int ph_idx = 0;                                   // phase counter for the trace hook
    const int n_ph = p.n_layers * 5;
    auto stamp = [&](int ev, bool who) {
        if (p.trace != nullptr && blk < p.trace_blocks && who && lane == 0)
            p.trace[((size_t)blk * n_ph + ph_idx) * 12 + ev] = wall_clock64();
    };
    // groups of this block in a phase with n groups: g = li * NB + blk
    auto ng_of = [&](int n) { return n > blk ? (n - blk + NB - 1) / NB : 0; };
    const int gph = half;
    const int n_g_qkv = (p.nq + 2 * p.nkv) * gph, n_g_h = (H + 1) / 2, n_g_f = (F + 1) / 2;
    const int ng_qkv = ng_of(n_g_qkv), ng_h = ng_of(n_g_h);
    // gate/up (54 % of the bytes): blocks with an odd (block % 8) — the odd XCDs under round-robin dispatch — stream ~7 % slower than the even
    // ones (measured with the trace hook: their arrival lags 4-5 us behind on a 34-us phase, every layer).  So every block takes cf = n / NB -
    // skew groups by the interleaved map and the remaining groups go to the even blocks only.
    const bool can_skew = (NB % 8) == 0;
    const int cf = can_skew ? (n_g_f / NB > p.skew_f ? n_g_f / NB - p.skew_f : 0) : 0;
    const bool fast_blk = can_skew && ((blk & 1) == 0);
    const int n_fast = NB / 2, fast_rank = (blk >> 3) * 4 + ((blk & 7) >> 1);
    const int rem_f = n_g_f - cf * NB;
    const int ng_f = can_skew ? cf + (fast_blk && rem_f > fast_rank ? (rem_f - fast_rank + n_fast - 1) / n_fast : 0) : ng_of(n_g_f);
    auto gid_f = [&](int li) { return !can_skew ? li * NB + blk : (li < cf ? li * NB + blk : cf * NB + (li - cf) * n_fast + fast_rank); };

    // sync wave: store `n` outq entries of `kind`, drain, then (grid) count the block and wait for everyone
    auto sync_flush = [&](int kind, int n, bf16_t* dst16, float* dst32, bf16_t* kc, bf16_t* vc, int N, bool barrier) {
        if (is_sync) {
            stamp(2, true);
#pragma unroll 1
            for (int e = lane; e < n; e += 64) {
                const uint32_t idx = outq[2 * e], val = outq[2 * e + 1];
                if (kind == DPK_PAIR) {
                    if ((int)idx + 1 < N) dp_st_u32((uint32_t*)(dst16 + idx), val);
                    else dp_st_u16(dst16 + idx, (bf16_t)(val & 0xffffu));
                } else if (kind == DPK_F32) {
                    if ((int)idx < N) dst32[idx] = __uint_as_float(val);
                } else {
                    const int head = (int)idx / hd, d = (int)idx % hd;
                    if (head < p.nq) dp_st_u16(dst16 + idx, (bf16_t)val);
                    else if (pos < p.max_ctx) {
                        const bool is_v = head >= p.nq + p.nkv;
                        const int kvh = is_v ? head - p.nq - p.nkv : head - p.nq;
                        dp_st_u16((is_v ? vc : kc) + ((int64_t)kvh * p.max_ctx + pos) * hd + d, (bf16_t)val);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            stamp(3, true);
            if (barrier) dp_arrive(sy, lane);
        }
    };
    auto sync_wait = [&]() {
        if (is_sync) { dp_wait(sy, lane); stamp(4, true); }
        ph_idx += 1;
    };

    // ---------------- phase bodies (workers) ----------------
    // batch shapes (rows x 16-B chunks per lane): q/k/v and o_proj row pairs 2 x 7 (K = hidden: a whole pair in one batch when K <= 3584),
    // gate/up row quads 4 x 4 (as the launch path's gemv_kernel<1,4>), down_proj row pairs 2 x 8 (K = intermediate).  16 slots = 64 VGPRs:
    // with 28 (quads 4 x 7) the allocator spilled inside the row loops; 7 workers x 16 KB in flight per CU is what the launch path averages
    static_assert(2 * 7 <= DP_SLOTS && 4 * 4 <= DP_SLOTS && 2 * 8 <= DP_SLOTS, "batch shapes fit the buffer");
    DpBufAll bq;
    DpBufAll& bg = bq;
    DpBufAll& bd = bq;
    const int nch_h = H >> 3, nch_q = QS >> 3, nch_f = F >> 3;

    auto rows_qkv = [&](const DpLayer& L, int li, const bf16_t* (&rows)[2], int (&ri)[2]) {
        const int gg = li * NB + blk;
        const int head = gg / gph, gi = gg % gph;
        if (head >= p.nq + p.nkv) { ri[0] = head * hd + gi * 2; ri[1] = ri[0] + 1; }
        else { ri[0] = head * hd + gi; ri[1] = ri[0] + half; }
        rows[0] = L.wqkv + (int64_t)ri[0] * H; rows[1] = L.wqkv + (int64_t)ri[1] * H;
    };
    auto rows_pair = [&](const bf16_t* W, int N, int K, int li, const bf16_t* (&rows)[2]) {
        const int n = (li * NB + blk) * 2;
        const int n1 = (n + 1 < N) ? n + 1 : n;
        rows[0] = W + (int64_t)n * K; rows[1] = W + (int64_t)n1 * K;
    };
    auto rows_gu = [&](const DpLayer& L, int li, const bf16_t* (&rows)[4]) {
        const int n = gid_f(li) * 2;
        const int n1 = (n + 1 < F) ? n + 1 : n;
        rows[0] = L.wg + (int64_t)n * H; rows[1] = L.wu + (int64_t)n * H;
        rows[2] = L.wg + (int64_t)n1 * H; rows[3] = L.wu + (int64_t)n1 * H;
    };
    // prefetch of the first batch of a phase (worker waves with at least one group there)
    auto pf_qkv = [&](const DpLayer& L, int lane) {
        dp_kill(bq);
        if (!is_sync && wave < ng_qkv) { const bf16_t* r[2]; int ri[2]; rows_qkv(L, wave, r, ri); dp_load_batch<2, 7>(bq, r, 0, lane, nch_h); }
    };
    auto pf_o = [&](const DpLayer& L, int lane) {
        dp_kill(bq);
        if (!is_sync && wave < ng_h) { const bf16_t* r[2]; rows_pair(L.wo, H, QS, wave, r); dp_load_batch<2, 7>(bq, r, 0, lane, nch_q); }
    };
    auto pf_gu = [&](const DpLayer& L, int lane) {
        dp_kill(bq);
        if (!is_sync && wave < ng_f) { const bf16_t* r[4]; rows_gu(L, wave, r); dp_load_batch<4, 4>(bg, r, 0, lane, nch_h); }
    };
    auto pf_dn = [&](const DpLayer& L, int lane) {
        dp_kill(bq);
        if (!is_sync && wave < ng_h) { const bf16_t* r[2]; rows_pair(L.wd, H, F, wave, r); dp_load_batch<2, 8>(bd, r, 0, lane, nch_f); }
    };
    bf16_t* cur = p.x0; bf16_t* nxt = p.x1;
    pf_qkv(p.layer[0], lane);
    dp_stage_x_norm(cur, p.layer[0].ln1, p.eps, H, sx, scratch, tid);

    // Order at the end of every phase:  rows done -> results in LDS -> block barrier -> sync wave: write-through stores, drain, ARRIVE ->
    // block barrier -> { workers: request the next phase's first batch | sync wave: WAIT } -> block barrier -> stage the activation.  The requests come AFTER barrier
    // A: a wave sits in its load instructions until the memory pipe has accepted them (~29 KB / us per CU), so a prefetch issued in front of
    // the barrier delayed the block's arrival by exactly the time it was supposed to hide (first version: workers done at 2 us, block at 7.8).
    for (int l = 0; l < p.n_layers; ++l) {
        const DpLayer& L = p.layer[l];
        const bool last = l + 1 == p.n_layers;
        // per-lane address arithmetic is loop-invariant: left alone, the compiler hoists ALL of it (a 64-bit offset pair per slot and phase) in
        // front of the layer loop and spills it (first build: 300 spills, reloads inside the row loops).  An opaque copy per phase keeps each
        // phase's addresses local to it.
        int lane_ = lane, tid_ = tid;
#define DP_OPAQUE() asm volatile("" : "+v"(lane_), "+v"(tid_))
        DP_OPAQUE();
        bf16_t* kc = p.kcache + (int64_t)l * p.kv_layer_stride;
        bf16_t* vc = p.vcache + (int64_t)l * p.kv_layer_stride;
        const __amdgpu_buffer_rsrc_t r_cur = dp_rsrc(cur, H * 2), r_nxt = dp_rsrc(nxt, H * 2);

        // ===== P1: q/k/v rows + bias + RoPE (gemv.hip qkv_decode_kernel) =====
        if (!is_sync) {
            int ri[2]; float e_b0 = 0.f, e_b1 = 0.f, e_c = 1.f, e_s = 0.f;
            stamp(0, wave == 0);
            dp_gemv_phase<2, 7>(bq, sx, nch_h, ng_qkv, wave, lane_,
                [&](int li, const bf16_t* (&rows)[2]) { int t[2]; rows_qkv(L, li, rows, t); },
                [&](int li) {
                    const bf16_t* r_[2]; rows_qkv(L, li, r_, ri);
                    if (lane_ < 2) {
                        const int gg = li * NB + blk, head = gg / gph, gi = gg % gph;
                        e_b0 = L.bqkv != nullptr ? bf2f(L.bqkv[ri[0]]) : 0.f;
                        e_b1 = L.bqkv != nullptr ? bf2f(L.bqkv[ri[1]]) : 0.f;
                        if (head < p.nq + p.nkv) { e_c = p.rope_cs[gi]; e_s = p.rope_cs[half + gi]; }
                    }
                },
                [&](int li, float (&acc)[2]) {
                    if (lane_ >= 2) return;
                    const int gg = li * NB + blk, head = gg / gph;
                    const bool is_v = head >= p.nq + p.nkv;
                    const float lo = bfround(acc[0] + e_b0), hi = bfround(acc[1] + e_b1);
                    float out = lane_ ? hi : lo;
                    if (!is_v) out = lane_ ? bfround(bfround(hi * e_c) + bfround(lo * e_s)) : bfround(bfround(lo * e_c) + bfround(-hi * e_s));
                    outq[(li * 2 + lane_) * 2] = (uint32_t)ri[lane_];
                    outq[(li * 2 + lane_) * 2 + 1] = (uint32_t)f2bf(out);
                });
            stamp(1, wave == 0); stamp(5 + wave, true);
        }
        DP_BB();
        sync_flush(DPK_QKV, ng_qkv * 2, p.q, nullptr, kc, vc, 0, true);
        DP_BB();
        pf_o(L, lane_);                                        // o_proj's rows stream under the barrier and the attention
        sync_wait();
        DP_BB();

        DP_OPAQUE();
        // ===== P2: attention over 256-key slices, one (head, slice) per block (gemv.hip attn_decode_head<true>) =====
        {
            const int item = blk, n_items = p.nq * n_active;
            const bool has = item < n_items;
            const int h = has ? item % p.nq : 0, slice = has ? item / p.nq : 0;
            const int kvh = h / (p.nq / p.nkv);
            const int key_lo = slice * 256;
            const int nkeys = nkeys_all < key_lo + 256 ? nkeys_all : key_lo + 256;
            stamp(0, wave == 0);
            if (has && tid_ < 128) sq[tid_] = dp_ld_bf16(dp_rsrc(p.q, QS * 2), h * 128 + tid_) * p.scale;
            DP_BB();
            if (has && !is_sync) {
                const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)(kc + (int64_t)kvh * p.max_ctx * 128), 0, p.max_ctx * 256, 0x00020000);
                const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)(vc + (int64_t)kvh * p.max_ctx * 128), 0, p.max_ctx * 256, 0x00020000);
                const int kq = lane_ >> 2, qd = lane_ & 3, sg = lane_ >> 4, dc = lane_ & 15;
                for (int ch = wave; ch < 16; ch += DPW) {
                    const int k0 = key_lo + ch * 16;
                    float m = -INFINITY, l_ = 0.f, o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = 0.f;
                    if (k0 < nkeys) {
                        u32x4 kk[4], vv[4];
                        const int key = k0 + kq;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            kk[j] = (key < nkeys) ? __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)(key * 256 + (qd * 32 + j * 8) * 2), 0, 16))
                                                  : (u32x4){0u, 0u, 0u, 0u};
                            const int vk = k0 + sg * 4 + j;
                            vv[j] = (vk < nkeys) ? __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rv, (unsigned)(vk * 256 + dc * 16), 0, 16))
                                                 : (u32x4){0u, 0u, 0u, 0u};
                        }
                        float a = 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 q0 = *(const f32x4*)(sq + qd * 32 + j * 8), q1 = *(const f32x4*)(sq + qd * 32 + j * 8 + 4);
                            a = fmaf(lo_bf(kk[j][0]), q0[0], a); a = fmaf(hi_bf(kk[j][0]), q0[1], a);
                            a = fmaf(lo_bf(kk[j][1]), q0[2], a); a = fmaf(hi_bf(kk[j][1]), q0[3], a);
                            a = fmaf(lo_bf(kk[j][2]), q1[0], a); a = fmaf(hi_bf(kk[j][2]), q1[1], a);
                            a = fmaf(lo_bf(kk[j][3]), q1[2], a); a = fmaf(hi_bf(kk[j][3]), q1[3], a);
                        }
                        a += __shfl_xor(a, 1, 64);
                        a += __shfl_xor(a, 2, 64);
                        const float s = (k0 + kq < nkeys) ? a : -INFINITY;
                        float cm = s;
                        cm = fmaxf(cm, __shfl_xor(cm, 4, 64)); cm = fmaxf(cm, __shfl_xor(cm, 8, 64));
                        cm = fmaxf(cm, __shfl_xor(cm, 16, 64)); cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
                        const float m_new = fmaxf(m, cm);
                        const float alpha = __expf(m - m_new);
                        const float pr = __expf(s - m_new);
                        float ps = pr;
                        ps += __shfl_xor(ps, 4, 64); ps += __shfl_xor(ps, 8, 64); ps += __shfl_xor(ps, 16, 64); ps += __shfl_xor(ps, 32, 64);
                        l_ = l_ * alpha + ps;
                        m = m_new;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] *= alpha;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float pj = __shfl(pr, (sg * 4 + j) * 4, 64);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                o[2 * e] = fmaf(pj, lo_bf(vv[j][e]), o[2 * e]);
                                o[2 * e + 1] = fmaf(pj, hi_bf(vv[j][e]), o[2 * e + 1]);
                            }
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) { o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64); }
                    if (lane_ < 16) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) so[ch * 128 + dc * 8 + e] = o[e];
                    }
                    if (lane_ == 0) { sml[ch * 2] = m; sml[ch * 2 + 1] = l_; }
                }
            }
            DP_BB();
            if (has && tid_ < 128) {
                float M = -INFINITY;
#pragma unroll
                for (int w_ = 0; w_ < 16; ++w_) M = fmaxf(M, sml[w_ * 2]);
                float Ls = 0.f, acc = 0.f;
#pragma unroll
                for (int w_ = 0; w_ < 16; ++w_) {
                    const float wgt = __expf(sml[w_ * 2] - M);
                    Ls += wgt * sml[w_ * 2 + 1];
                    acc += wgt * so[w_ * 128 + tid_];
                }
                fin[tid_] = acc;
                if (tid_ == 0) { fin[128] = M; fin[129] = Ls; }
            }
            stamp(1, wave == 0); stamp(5 + wave, !is_sync);
            DP_BB();
            if (is_sync) {
                stamp(2, true);
                if (has) {
                    const int64_t slot = (int64_t)slice * p.nq + h;
                    dp_st_f2(p.part_o + slot * 128 + 2 * lane_, fin[2 * lane_], fin[2 * lane_ + 1]);
                    if (lane_ == 0) dp_st_f2(p.part_ml + slot * 2, fin[128], fin[129]);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                stamp(3, true);
                dp_arrive(sy, lane_);
                dp_wait(sy, lane_);
                stamp(4, true);
            }
            ph_idx += 1;
            DP_BB();
        }

        DP_OPAQUE();
        // ===== P3: o_proj over the merged attention output + residual (gemv_kernel<2,7>) =====
        dp_stage_x_attn(p.part_o, p.part_ml, n_active, p.nq, sx, wsm, tid_);
        if (!is_sync) {
            float e_res = 0.f;
            stamp(0, wave == 0);
            dp_gemv_phase<2, 7>(bq, sx, nch_q, ng_h, wave, lane_,
                [&](int li, const bf16_t* (&rows)[2]) { rows_pair(L.wo, H, QS, li, rows); },
                [&](int li) { const int nn = (li * NB + blk) * 2 + lane_; e_res = (lane_ < 2 && nn < H) ? dp_ld_bf16(r_cur, nn) : 0.f; },
                [&](int li, float (&acc)[2]) {
                    const float v = bfround(lane_ == 0 ? acc[0] : acc[1]) + e_res;
                    const bf16_t o = f2bf(v);
                    const bf16_t o1 = (bf16_t)__shfl((int)o, 1, 64);
                    if (lane_ == 0) { outq[li * 2] = (uint32_t)((li * NB + blk) * 2); outq[li * 2 + 1] = (uint32_t)o | ((uint32_t)o1 << 16); }
                });
            stamp(1, wave == 0); stamp(5 + wave, true);
        }
        DP_BB();
        sync_flush(DPK_PAIR, ng_h, nxt, nullptr, nullptr, nullptr, H, true);
        DP_BB();
        pf_gu(L, lane_);
        sync_wait();
        DP_BB();

        DP_OPAQUE();
        // ===== P4: RMSNorm + gate/up + silu*mul (gemv_kernel<1,4>) =====
        dp_stage_x_norm(nxt, L.ln2, p.eps, H, sx, scratch, tid_);
        if (!is_sync) {
            stamp(0, wave == 0);
            dp_gemv_phase<4, 4>(bg, sx, nch_h, ng_f, wave, lane_,
                [&](int li, const bf16_t* (&rows)[4]) { rows_gu(L, li, rows); },
                [&](int) {},
                [&](int li, float (&acc)[4]) {
                    const float gv = bfround(lane_ == 0 ? acc[0] : acc[2]), uv = bfround(lane_ == 0 ? acc[1] : acc[3]);
                    const bf16_t o = f2bf(bfround(silu_f(gv)) * uv);
                    const bf16_t o1 = (bf16_t)__shfl((int)o, 1, 64);
                    if (lane_ == 0) { outq[li * 2] = (uint32_t)(gid_f(li) * 2); outq[li * 2 + 1] = (uint32_t)o | ((uint32_t)o1 << 16); }
                });
            stamp(1, wave == 0); stamp(5 + wave, true);
        }
        DP_BB();
        sync_flush(DPK_PAIR, ng_f, p.act, nullptr, nullptr, nullptr, F, true);
        DP_BB();
        pf_dn(L, lane_);
        sync_wait();
        DP_BB();

        DP_OPAQUE();
        // ===== P5: down_proj + residual (gemv_kernel<0,4>) =====
        dp_stage_x_copy(p.act, F, sx, tid_);
        if (!is_sync) {
            float e_res = 0.f;
            stamp(0, wave == 0);
            dp_gemv_phase<2, 8>(bd, sx, nch_f, ng_h, wave, lane_,
                [&](int li, const bf16_t* (&rows)[2]) { rows_pair(L.wd, H, F, li, rows); },
                [&](int li) { const int nn = (li * NB + blk) * 2 + lane_; e_res = (lane_ < 2 && nn < H) ? dp_ld_bf16(r_nxt, nn) : 0.f; },
                [&](int li, float (&acc)[2]) {
                    const float v = bfround(lane_ == 0 ? acc[0] : acc[1]) + e_res;
                    const bf16_t o = f2bf(v);
                    const bf16_t o1 = (bf16_t)__shfl((int)o, 1, 64);
                    if (lane_ == 0) { outq[li * 2] = (uint32_t)((li * NB + blk) * 2); outq[li * 2 + 1] = (uint32_t)o | ((uint32_t)o1 << 16); }
                });
            stamp(1, wave == 0); stamp(5 + wave, true);
        }
        DP_BB();
        // the last layer's output leaves at the end of the launch: the head (RMSNorm + lm_head rows) is the next kernel of the stream
        sync_flush(DPK_PAIR, ng_h, cur, nullptr, nullptr, nullptr, H, !last);
        if (last) break;
        DP_BB();
        pf_qkv(p.layer[l + 1], lane_);
        sync_wait();
        DP_BB();
        dp_stage_x_norm(cur, p.layer[l + 1].ln1, p.eps, H, sx, scratch, tid_);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static int dp_cu_count() {
    static thread_local int per_dev[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& n = per_dev[dev & 15];
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}
int decode_persist_blocks() { return dp_cu_count(); }

bool decode_persist_supported(int H, int F, int nq, int nkv, int hd, int n_layers, int max_ctx, int vocab) {
    const int NB = dp_cu_count();
    return hd == 128 && H % 8 == 0 && F % 8 == 0 && H >= 8 && H <= 8192 && n_layers >= 1 && n_layers <= DP_MAX_LAYERS && nkv > 0 && nq % nkv == 0 &&
           max_ctx <= 2048 && nq * cdiv(max_ctx, 256) <= NB && vocab >= 4;
}

static unsigned long long* g_dp_trace = nullptr;
static int g_dp_trace_blocks = 0;
void decode_persist_set_trace(unsigned long long* buf, int n_blocks) { g_dp_trace = buf; g_dp_trace_blocks = n_blocks; }

int launch_decode_persist(DpArgs& a, hipStream_t s) {
    const int NB = dp_cu_count();
    a.trace = g_dp_trace; a.trace_blocks = g_dp_trace_blocks;
    static int skew = -1;
    if (skew < 0) { const char* e = getenv("VILA_DECODE_PERSIST_SKEW"); skew = (e && e[0] >= '0' && e[0] <= '9') ? atoi(e) : 2; }
    a.skew_f = skew;
    const int QS = a.nq * a.hd;
    int kmax = a.H > a.F ? a.H : a.F; if (QS > kmax) kmax = QS;
    size_t off = align_up((size_t)kmax * 2, 256);
    a.lds_scratch = (int)off; off += 64;
    const int e_qkv = cdiv((a.nq + 2 * a.nkv) * (a.hd / 2), NB) * 2, e_h = cdiv((a.H + 1) / 2, NB), e_f = cdiv((a.F + 1) / 2, NB) + a.skew_f + 2;
    int emax = e_qkv; if (e_h > emax) emax = e_h; if (e_f > emax) emax = e_f;
    a.lds_outq = (int)off; off += align_up((size_t)emax * 8, 256);
    a.lds_attn = (int)off; off += (128 + 16 * 128 + 32 + 132) * 4;
    off = align_up(off, 256);
    a.lds_wsm = (int)off; off += align_up((size_t)cdiv(a.max_ctx, 256) * a.nq * 4, 256);
    VILA_REQUIRE(off <= 160 * 1024, "decode_persist: %zu B of LDS needed", off);
    static thread_local size_t attr_bytes[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_bytes[dev & 15] < off) {
        VILA_HIP(hipFuncSetAttribute((const void*)decode_token_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)off));
        attr_bytes[dev & 15] = off;
    }
    hipLaunchKernelGGL(decode_token_kernel, dim3(NB), dim3(DP_THREADS), off, s, a);
    VILA_LAUNCH_CHECK();
    return 0;
}
