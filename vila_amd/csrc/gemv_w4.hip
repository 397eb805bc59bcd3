// W4A16 decode GEMVs (SURVEY.md §8f row 3 / BASELINE configs[4]: "AWQ int4 dequant-GEMV").  The reference names TinyChat
// (mit-han-lab/llm-awq, external, un-pinned, zero call sites in-tree: README.md:87) as its W4A16 backend; there is no reference
// code for it, so the format below is ours and parity is against a CPU dequantise-then-fp32 oracle of the SAME quantised weights.
//
// Format (AWQ-style asymmetric uint4, groups of 128 along K):
//   Wq  [N][K/8]   u32: 8 nibbles; nibble j (j<4) = element 2j, nibble j+4 = element 2j+1 of the 8-element run, so that
//                  ((w >> 4j) & 0x000F000F) | 0x43004300 is the bf16 PAIR (128+q[2j], 128+q[2j+1]) matching the bf16 x pair
//   Wsz [N][K/128] u32: lo = bf16 scale, hi = bf16 (128 + zero)       (dequant: (q - zero) * scale)
// Per 16-B load a lane covers 32 weights: dot = sum x_k (128+q_k) (32 FMAs, 3 VALU ops per weight incl. unpack), then
//   acc += scale * (dot - (128+zero) * xs[chunk])   with xs[chunk] = sum of the 32 activations, precomputed once per block in LDS.
// HBM-bound like the bf16 GEMVs (0.53 B per weight); VALU budget ~100 lane-ops per 16 B = ~50 % of the CU at full HBM rate.
#include "gemv_common.h"
#include "w4.h"

#define W4_U 4

__device__ __forceinline__ float dot32_w4(const u32x4 wq, const u32x4 (&xv)[4]) {
    float d = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = ((wq[w] >> (4 * j)) & 0x000F000Fu) | 0x43004300u;
            const uint32_t xp = xv[w][j];
            d = fmaf(lo_bf(t), lo_bf(xp), d);
            d = fmaf(hi_bf(t), hi_bf(xp), d);
        }
    }
    return d;
}

// R rows of packed weights x the staged activation; returns wave-reduced sums
template <int R>
__device__ __forceinline__ void w4_rows_dot(const uint32_t* const (&wq)[R], const uint32_t* const (&wsz)[R], const bf16_t* sx, const float* xs,
                                            int K, int lane, float (&acc)[R]) {
    const int nchunk = K >> 5;                      // 32-weight chunks (16 B of nibbles)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int c0 = 0; c0 < nchunk; c0 += 64 * W4_U) {
        u32x4 wv[W4_U][R];
        uint32_t sz[W4_U][R];
#pragma unroll
        for (int u = 0; u < W4_U; ++u) {
            const int c = c0 + u * 64 + lane;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                wv[u][r] = (c < nchunk) ? __builtin_nontemporal_load((const u32x4*)(wq[r] + c * 4)) : (u32x4){0u, 0u, 0u, 0u};
                sz[u][r] = (c < nchunk) ? wsz[r][c >> 2] : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < W4_U; ++u) {
            const int c = c0 + u * 64 + lane;
            if (c < nchunk) {
                u32x4 xv[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) xv[w] = *(const u32x4*)(sx + c * 32 + w * 8);
                const float xsum = xs[c];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float d = dot32_w4(wv[u][r], xv);
                    acc[r] = fmaf(lo_bf(sz[u][r]), d - hi_bf(sz[u][r]) * xsum, acc[r]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = wave_sum(acc[r]);
}

// MODE 0: y = W x (+bias)(+residual) ; 1: y = silu(Wg x) * (Wu x) ; 3: fused QKV + bias + RoPE + KV append
template <int MODE>
__global__ __launch_bounds__(256) void gemv_w4_kernel(GemvW4Args p, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* sx = (bf16_t*)smem;
    float* xs = (float*)(smem + ((p.K * 2 + 15) & ~15));
    float* scratch = xs + (p.K >> 5);
    stage_x(p.x, p.norm_w, p.eps, p.K, sx, scratch);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < (p.K >> 5); c += 256) {       // per-chunk activation sums (the zero-point correction term)
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const u32x4 v = *(const u32x4*)(sx + c * 32 + w * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) s += lo_bf(v[k]) + hi_bf(v[k]);
        }
        xs[c] = s;
    }
    __syncthreads();
    const int64_t rq = p.K >> 3, rs = p.K >> 7;        // row strides (u32 words) of Wq / Wsz
    const int half = p.hd >> 1;
    for (int g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
        int r0, r1;
        const uint32_t* wq[2];
        const uint32_t* wsz[2];
        int head = 0, gi = 0; bool is_v = false;
        if (MODE == 1) {
            r0 = r1 = g;
            wq[0] = p.Wq + r0 * rq; wsz[0] = p.Wsz + r0 * rs; wq[1] = p.Wq2 + r0 * rq; wsz[1] = p.Wsz2 + r0 * rs;
        } else {
            if (MODE == 3) {
                head = g / half; gi = g % half; is_v = head >= p.nq + p.nkv;
                if (is_v) { r0 = head * p.hd + gi * 2; r1 = r0 + 1; } else { r0 = head * p.hd + gi; r1 = r0 + half; }
            } else {
                r0 = g * 2; r1 = (r0 + 1 < p.N) ? r0 + 1 : r0;
            }
            wq[0] = p.Wq + r0 * rq; wsz[0] = p.Wsz + r0 * rs; wq[1] = p.Wq + r1 * rq; wsz[1] = p.Wsz + r1 * rs;
        }
        float acc[2];
        w4_rows_dot<2>(wq, wsz, sx, xs, p.K, lane, acc);
        if (MODE == 1) {
            if (lane == 0) p.y[g] = f2bf(bfround(silu_f(bfround(acc[0]))) * bfround(acc[1]));
        } else if (MODE == 0) {
            if (lane < 2 && g * 2 + lane < p.N) {
                const int nn = g * 2 + lane;
                float v = lane ? acc[1] : acc[0];
                if (p.bias != nullptr) v += bf2f(p.bias[nn]);
                if (p.residual != nullptr) v = bfround(v) + bf2f(p.residual[nn]);
                p.y[nn] = f2bf(v);
            }
        } else {
            if (lane < 2) {
                const int pos = *p.pos_ptr;
                const float lo = bfround(acc[0] + (p.bias != nullptr ? bf2f(p.bias[r0]) : 0.f));
                const float hi = bfround(acc[1] + (p.bias != nullptr ? bf2f(p.bias[r1]) : 0.f));
                float out = lane ? hi : lo;
                if (!is_v) {
                    const float c = p.rope_cs[gi], sn = p.rope_cs[half + gi];
                    out = lane ? bfround(bfround(hi * c) + bfround(lo * sn)) : bfround(bfround(lo * c) + bfround(-hi * sn));
                }
                const int row = lane ? r1 : r0;
                if (head < p.nq) {
                    p.q_out[row] = f2bf(out);
                } else if (pos < p.max_ctx) {
                    const int kvh = is_v ? head - p.nq - p.nkv : head - p.nq;
                    bf16_t* dst = (is_v ? p.vcache : p.kcache) + ((int64_t)kvh * p.max_ctx + pos) * p.hd;
                    dst[row - head * p.hd] = f2bf(out);
                }
            }
        }
    }
}

static inline int w4_grid(int n_groups) {
    int want = cdiv(n_groups, 4);
    if (want > 1024) want = 1024;
    return want <= 256 ? want : cdiv(want, 256) * 256;
}

int launch_gemv_w4(const GemvW4Args& a, hipStream_t s) {
    VILA_REQUIRE(a.K % 128 == 0 && a.K > 0, "gemv_w4: K=%d must be a multiple of the 128-wide quantisation group", a.K);
    VILA_REQUIRE((uintptr_t)a.Wq % 16 == 0 && (uintptr_t)a.x % 16 == 0, "gemv_w4: pointer alignment");
    const size_t lds = ((size_t)a.K * 2 + 15) / 16 * 16 + (size_t)(a.K / 32) * 4 + 32;
    if (a.mode == 1) {
        VILA_REQUIRE(a.Wq2 != nullptr && a.Wsz2 != nullptr, "gemv_w4: gate/up mode needs the up matrix");
        hipLaunchKernelGGL(gemv_w4_kernel<1>, dim3(w4_grid(a.N)), dim3(256), lds, s, a, a.N);
    } else if (a.mode == 3) {
        const int n_groups = (a.nq + 2 * a.nkv) * (a.hd / 2);
        VILA_REQUIRE(a.rope_cs != nullptr && a.pos_ptr != nullptr, "gemv_w4: qkv mode needs rope table and position");
        hipLaunchKernelGGL(gemv_w4_kernel<3>, dim3(w4_grid(n_groups)), dim3(256), lds, s, a, n_groups);
    } else {
        hipLaunchKernelGGL(gemv_w4_kernel<0>, dim3(w4_grid(cdiv(a.N, 2))), dim3(256), lds, s, a, cdiv(a.N, 2));
    }
    VILA_LAUNCH_CHECK();
    return 0;
}
