// HBM-bound helpers around the MFMA kernels: norms, patchify, space-to-depth, RoPE + KV scatter, row gathers, argmax.
// All are vectorised 16 B per lane (guide G13) and keep statistics in fp32.
#include <stdlib.h>
#include "kernels.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm (SigLIP layer_norm1/2 eps 1e-6, modeling_siglip.py:723-725; projector nn.LayerNorm eps 1e-5,
// base_projector.py:147): y = bf16((x-mean)*rstd*w + b), two-pass variance in fp32.  One block per row.
// RMSNorm (Qwen2RMSNorm): y = bf16(w * bf16(x * rsqrt(mean(x^2)+eps)))  -- the double rounding is HF's.
// ------------------------------------------------------------------------------------------------
template <bool RMS>
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                   const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int cols, float eps) {
    __shared__ float scratch[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const bf16_t* xr = x + (int64_t)row * cols;
    bf16_t* yr = y + (int64_t)row * cols;
    const int nch = cols >> 3;
    constexpr int MAXC = 8;  // cols <= 16384
    u32x4 v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + 256 * i;
        v[i] = (u32x4){0u, 0u, 0u, 0u};
        if (c < nch) {
            v[i] = *(const u32x4*)(xr + c * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a = lo_bf(v[i][k]), bb = hi_bf(v[i][k]);
                s += RMS ? (a * a + bb * bb) : (a + bb);
            }
        }
    }
    s = block_sum_256(s, scratch);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / cols + eps);
    } else {
        mean = s / cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = tid + 256 * i;
            if (c < nch) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = lo_bf(v[i][k]) - mean, bb = hi_bf(v[i][k]) - mean;
                    q += a * a + bb * bb;
                }
            }
        }
        q = block_sum_256(q, scratch);
        rstd = rsqrtf(q / cols + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + 256 * i;
        if (c < nch) {
            const u32x4 wv = *(const u32x4*)(w + c * 8);
            u32x4 bv = (u32x4){0u, 0u, 0u, 0u};
            if (!RMS && b != nullptr) bv = *(const u32x4*)(b + c * 8);
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = lo_bf(v[i][k]), bb = hi_bf(v[i][k]);
                if (RMS) {
                    a = lo_bf(wv[k]) * bfround(a * rstd);
                    bb = hi_bf(wv[k]) * bfround(bb * rstd);
                } else {
                    a = (a - mean) * rstd * lo_bf(wv[k]) + lo_bf(bv[k]);
                    bb = (bb - mean) * rstd * hi_bf(wv[k]) + hi_bf(bv[k]);
                }
                o[k] = pack2bf(a, bb);
            }
            *(u32x4*)(yr + c * 8) = o;
        }
    }
}

// Rows of up to 1536 columns (the tower's LayerNorms): ONE WAVE per row, no LDS, no barriers — x, w and b are all requested
// before anything is reduced, the two reductions are wave shuffles.  The block-per-row kernel above spends its time in a dependent chain
// (load x -> block reduce -> block reduce -> load w, b -> store) that costs 8 us for 2.4 MB at M = 1024 x 1152; the chain here is
// load -> shuffle reduce (x2) -> store.  Same arithmetic per element; the fp32 sums are taken in another order.
template <bool RMS, int MAXC>
__global__ __launch_bounds__(256) void norm_wave_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                        bf16_t* __restrict__ y, int rows, int cols, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;                                     // wave-uniform
    const bf16_t* xr = x + (int64_t)row * cols;
    bf16_t* yr = y + (int64_t)row * cols;
    const int nch = cols >> 3;                                   // <= 64 * MAXC chunks of 16 B
    u32x4 v[MAXC], wv[MAXC], bv[MAXC];
    // every load is UNCONDITIONAL (chunks beyond the row re-read its last chunk and are zeroed by a select): a branch around a load makes
    // the compiler drain the memory queue at the join, which serialised the 14-21 loads of a 3584-wide row (18.7 us instead of 6)
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i, cc = c < nch ? c : nch - 1;
        v[i] = *(const u32x4*)(xr + cc * 8);
        wv[i] = *(const u32x4*)(w + cc * 8);
        if constexpr (!RMS) bv[i] = *(const u32x4*)((b != nullptr ? b : w) + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const bool ok = lane + 64 * i < nch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[i][k] = ok ? v[i][k] : 0u;
            if constexpr (!RMS) bv[i][k] = (ok && b != nullptr) ? bv[i][k] : 0u; else bv[i][k] = 0u;
        }
    }
    auto wave_sum = [](float t) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        return t;
    };
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = lo_bf(v[i][k]), bb = hi_bf(v[i][k]);
            s += RMS ? (a * a + bb * bb) : (a + bb);
        }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / cols + eps);
    } else {
        mean = s / cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            if (lane + 64 * i < nch) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = lo_bf(v[i][k]) - mean, bb = hi_bf(v[i][k]) - mean;
                    q += a * a + bb * bb;
                }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / cols + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = lo_bf(v[i][k]), bb = hi_bf(v[i][k]);
                if (RMS) {
                    a = lo_bf(wv[i][k]) * bfround(a * rstd);
                    bb = hi_bf(wv[i][k]) * bfround(bb * rstd);
                } else {
                    a = (a - mean) * rstd * lo_bf(wv[i][k]) + lo_bf(bv[i][k]);
                    bb = (bb - mean) * rstd * hi_bf(wv[i][k]) + hi_bf(bv[i][k]);
                }
                o[k] = pack2bf(a, bb);
            }
            *(u32x4*)(yr + c * 8) = o;
        }
    }
}

// norm_kernel with every load requested UP FRONT ("LAT", added at the end of round 4 from the ISA alone, OFF by default until measured —
// VILA_NORM_LAT=1): norm_kernel's loads sit inside `if (c < nch)` (x) and behind the block reduction (w, b), so a 3584-wide row is
// x -> x -> reduce -> w -> w = five dependent round trips (8 `global_load -> s_waitcnt vmcnt(0)` pairs in the RMS kernel's ISA); one block
// per row means nothing else hides them when the rows are few (8-16 rows in the batched decode step: 2 x 5.1 us per layer) and the kernel
// sits at 7.8 us for 11 MB at S = 769.  Here x, w (and b) are requested unconditionally (chunks past the row re-read its last chunk and are
// zeroed by a select) before anything is reduced.  Same per-thread summation order and the same block reduction: bit-identical results.
template <bool RMS, int MAXC>
__global__ __launch_bounds__(256) void norm_block_lat_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                             bf16_t* __restrict__ y, int cols, float eps) {
    __shared__ float scratch[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const bf16_t* xr = x + (int64_t)row * cols;
    bf16_t* yr = y + (int64_t)row * cols;
    const int nch = cols >> 3;                                   // <= 256 * MAXC
    u32x4 v[MAXC], wv[MAXC], bv[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + 256 * i, cc = c < nch ? c : nch - 1;
        v[i] = *(const u32x4*)(xr + cc * 8);
        wv[i] = *(const u32x4*)(w + cc * 8);
        if constexpr (!RMS) bv[i] = *(const u32x4*)((b != nullptr ? b : w) + cc * 8);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const bool ok = tid + 256 * i < nch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[i][k] = ok ? v[i][k] : 0u;
            if constexpr (!RMS) bv[i][k] = (ok && b != nullptr) ? bv[i][k] : 0u; else bv[i][k] = 0u;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = lo_bf(v[i][k]), bb = hi_bf(v[i][k]);
            s += RMS ? (a * a + bb * bb) : (a + bb);
        }
    s = block_sum_256(s, scratch);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / cols + eps);
    } else {
        mean = s / cols;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            if (tid + 256 * i < nch) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a = lo_bf(v[i][k]) - mean, bb = hi_bf(v[i][k]) - mean;
                    q += a * a + bb * bb;
                }
            }
        }
        q = block_sum_256(q, scratch);
        rstd = rsqrtf(q / cols + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = tid + 256 * i;
        if (c < nch) {
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a = lo_bf(v[i][k]), bb = hi_bf(v[i][k]);
                if (RMS) {
                    a = lo_bf(wv[i][k]) * bfround(a * rstd);
                    bb = hi_bf(wv[i][k]) * bfround(bb * rstd);
                } else {
                    a = (a - mean) * rstd * lo_bf(wv[i][k]) + lo_bf(bv[i][k]);
                    bb = (bb - mean) * rstd * hi_bf(wv[i][k]) + hi_bf(bv[i][k]);
                }
                o[k] = pack2bf(a, bb);
            }
            *(u32x4*)(yr + c * 8) = o;
        }
    }
}
static int g_norm_lat = -1;               // -1 = VILA_NORM_LAT from the environment (default 1 since round 5: bit-identical outputs, batch-8 decode
                                          // 3.551 -> 3.497 ms per step, profiles/r05_second_call_ab.log), 0 / 1 = forced (vila_norm_force_lat)
extern "C" void vila_norm_force_lat(int on) { g_norm_lat = on; }
static int norm_lat() {
    if (g_norm_lat < 0) { const char* e = getenv("VILA_NORM_LAT"); g_norm_lat = (e && e[0] == '0') ? 0 : 1; }
    return g_norm_lat;
}
template <bool RMS>
static int launch_norm_block_lat(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int rows, int cols, float eps, hipStream_t s) {
    if (cols <= 4096) hipLaunchKernelGGL((norm_block_lat_kernel<RMS, 2>), dim3(rows), dim3(256), 0, s, x, w, b, y, cols, eps);
    else if (cols <= 8192) hipLaunchKernelGGL((norm_block_lat_kernel<RMS, 4>), dim3(rows), dim3(256), 0, s, x, w, b, y, cols, eps);
    else hipLaunchKernelGGL((norm_block_lat_kernel<RMS, 8>), dim3(rows), dim3(256), 0, s, x, w, b, y, cols, eps);
    VILA_LAUNCH_CHECK();
    return 0;
}

int launch_layernorm(const bf16_t* x, const bf16_t* w, const bf16_t* b, bf16_t* y, int rows, int cols, float eps, hipStream_t s) {
    VILA_REQUIRE(cols % 8 == 0 && cols <= 16384 && rows > 0, "layernorm: cols=%d must be a multiple of 8 and <= 16384", cols);
    if (cols <= 1536) {          // wider rows: the block-per-row kernel wins (3584 columns: 6.7 us vs 19 for one wave per 7-KB row at S = 769)
        hipLaunchKernelGGL((norm_wave_kernel<false, 3>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, w, b, y, rows, cols, eps);
        VILA_LAUNCH_CHECK();
        return 0;
    }
    if (norm_lat()) return launch_norm_block_lat<false>(x, w, b, y, rows, cols, eps, s);
    hipLaunchKernelGGL(norm_kernel<false>, dim3(rows), dim3(256), 0, s, x, w, b, y, cols, eps);
    VILA_LAUNCH_CHECK();
    return 0;
}
int launch_rmsnorm(const bf16_t* x, const bf16_t* w, bf16_t* y, int rows, int cols, float eps, hipStream_t s) {
    VILA_REQUIRE(cols % 8 == 0 && cols <= 16384 && rows > 0, "rmsnorm: cols=%d must be a multiple of 8 and <= 16384", cols);
    if (cols <= 1536) {
        hipLaunchKernelGGL((norm_wave_kernel<true, 3>), dim3(cdiv(rows, 4)), dim3(256), 0, s, x, w, (const bf16_t*)nullptr, y, rows, cols, eps);
        VILA_LAUNCH_CHECK();
        return 0;
    }
    if (norm_lat()) return launch_norm_block_lat<true>(x, w, nullptr, y, rows, cols, eps, s);
    hipLaunchKernelGGL(norm_kernel<true>, dim3(rows), dim3(256), 0, s, x, w, (const bf16_t*)nullptr, y, cols, eps);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// im2col for the patch-embed Conv2d(k=s=P, valid) (modeling_siglip.py:269-275,320-323):
// out[(b*gh+gy)*gw+gx][c*P*P + ky*P + kx] = px[b][c][gy*P+ky][gx*P+kx], zero-padded to Kp columns.
// One thread per (token, c, ky): reads P contiguous pixels of one image row (28 B, coalesced across gx).
// ------------------------------------------------------------------------------------------------
__global__ void im2col_kernel(const bf16_t* __restrict__ px, bf16_t* __restrict__ out, int B, int C, int H, int W, int P, int Kp) {
    const int gh = H / P, gw = W / P;
    const int64_t total = (int64_t)B * gh * C * P * gw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // order: gx fastest so that a wave walks along one pixel row
        int64_t r = i;
        const int gx = r % gw; r /= gw;
        const int ky = r % P; r /= P;
        const int c = r % C; r /= C;
        const int gy = r % gh; r /= gh;
        const int b = (int)r;
        const bf16_t* src = px + (((int64_t)b * C + c) * H + gy * P + ky) * W + gx * P;
        bf16_t* dst = out + ((int64_t)(b * gh + gy) * gw + gx) * Kp + (c * P + ky) * P;
        for (int kx = 0; kx < P; ++kx) dst[kx] = src[kx];
    }
}
__global__ void zero_tail_kernel(bf16_t* __restrict__ out, int rows, int K, int Kp) {
    const int pad = Kp - K;
    const int64_t total = (int64_t)rows * pad;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        out[(i / pad) * Kp + K + (i % pad)] = 0;
}
int launch_im2col(const bf16_t* px, bf16_t* out, int B, int C, int H, int W, int P, int Kp, hipStream_t s) {
    VILA_REQUIRE(H % P == 0 && W % P == 0, "im2col: image %dx%d is not a multiple of the patch size %d", H, W, P);
    const int K = C * P * P;
    VILA_REQUIRE(Kp >= K, "im2col: Kp < K");
    const int64_t total = (int64_t)B * (H / P) * C * P * (W / P);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, s, px, out, B, C, H, W, P, Kp);
    VILA_LAUNCH_CHECK();
    if (Kp > K) {
        const int rows = B * (H / P) * (W / P);
        hipLaunchKernelGGL(zero_tail_kernel, dim3(cdiv(rows * (Kp - K), 256)), dim3(256), 0, s, out, rows, K, Kp);
        VILA_LAUNCH_CHECK();
    }
    return 0;
}

__global__ void pad_rows_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int rows, int K, int Kp) {
    const int64_t total = (int64_t)rows * Kp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / Kp), c = (int)(i % Kp);
        out[i] = c < K ? in[(int64_t)r * K + c] : (bf16_t)0;
    }
}
int launch_pad_rows(const bf16_t* in, bf16_t* out, int rows, int K, int Kp, hipStream_t s) {
    const int64_t total = (int64_t)rows * Kp;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, s, in, out, rows, K, Kp);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// space-to-depth = DownSampleBlock.flat_square / flat_square_2x2 / flat_square_3x3 (base_projector.py:58-123):
// y[b][i*gd+j][(a*k+bb)*C + ch] = x[b][(k*i+a)*g + (k*j+bb)][ch], zero where k*i+a >= g or k*j+bb >= g.
// ------------------------------------------------------------------------------------------------
__global__ void s2d_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B, int g, int C, int k) {
    const int gd = (g + k - 1) / k;
    const int c8 = C >> 3;
    const int64_t total = (int64_t)B * gd * gd * k * k * c8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int ch = (int)(r % c8); r /= c8;
        const int bb = (int)(r % k); r /= k;
        const int a = (int)(r % k); r /= k;
        const int j = (int)(r % gd); r /= gd;
        const int ii = (int)(r % gd); r /= gd;
        const int b = (int)r;
        const int sy = k * ii + a, sx = k * j + bb;
        u32x4 v = (u32x4){0u, 0u, 0u, 0u};
        if (sy < g && sx < g) v = *(const u32x4*)(x + ((int64_t)b * g * g + sy * g + sx) * C + ch * 8);
        *(u32x4*)(y + (((int64_t)b * gd * gd + ii * gd + j) * k * k + a * k + bb) * C + ch * 8) = v;
    }
}
int launch_space_to_depth(const bf16_t* x, bf16_t* y, int B, int g, int C, int k, hipStream_t s) {
    VILA_REQUIRE(C % 8 == 0 && (k == 2 || k == 3), "space_to_depth: C=%d must be a multiple of 8, k=%d in {2,3}", C, k);
    const int gd = (g + k - 1) / k;
    const int64_t total = (int64_t)B * gd * gd * k * k * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(s2d_kernel, dim3(grid), dim3(256), 0, s, x, y, B, g, C, k);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// RoPE (HF Qwen2RotaryEmbedding + apply_rotary_pos_emb, rotate-half): inv_freq = theta^(-2i/hd), angle = pos*inv_freq
// in fp32, cos/sin CAST TO THE ACTIVATION DTYPE (bf16) before use; q' = bf16(bf16(q*cos) + bf16(rot(q)*sin)).
// ------------------------------------------------------------------------------------------------
__global__ void rope_table_kernel(const int32_t* __restrict__ pos, float* __restrict__ cs, float* __restrict__ sn, int S, int hd, float theta) {
    const int half = hd >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * half) return;
    const int s = i / half, d = i % half;
    const float inv = 1.0f / powf(theta, (float)(2 * d) / (float)hd);
    const float ang = (float)pos[s] * inv;
    cs[i] = bfround(cosf(ang));
    sn[i] = bfround(sinf(ang));
}
int launch_rope_table(const int32_t* pos, float* cs, float* sn, int S, int hd, float theta, hipStream_t s) {
    hipLaunchKernelGGL(rope_table_kernel, dim3(cdiv(S * hd / 2, 256)), dim3(256), 0, s, pos, cs, sn, S, hd, theta);
    VILA_LAUNCH_CHECK();
    return 0;
}

// one thread per (token, head, 8-wide d chunk of the low half); heads [0,nq) are q, [nq,nq+nkv) k, then v copy
__global__ void rope_kv_kernel(bf16_t* __restrict__ qkv, const float* __restrict__ cs, const float* __restrict__ sn,
                               const int32_t* __restrict__ pos, const int32_t* __restrict__ seq_of_tok,
                               bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache,
                               int S, int nq, int nkv, int hd, int max_ctx) {
    const int half = hd >> 1, cpr = half >> 3;           // chunks per half row
    const int heads = nq + 2 * nkv;
    const int64_t total = (int64_t)S * heads * cpr;
    const int row = (nq + 2 * nkv) * hd;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int ch = (int)(r % cpr); r /= cpr;
        const int hh = (int)(r % heads); r /= heads;
        const int s = (int)r;
        bf16_t* base = qkv + (int64_t)s * row + hh * hd + ch * 8;
        u32x4 x1 = *(const u32x4*)base;
        u32x4 x2 = *(const u32x4*)(base + half);
        const int p = pos[s];
        const int sq = seq_of_tok != nullptr ? seq_of_tok[s] : 0;
        if (hh < nq + nkv) {
            const float* c = cs + (int64_t)s * half + ch * 8;
            const float* sv = sn + (int64_t)s * half + ch * 8;
            u32x4 o1, o2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a0 = lo_bf(x1[k]), a1 = hi_bf(x1[k]), b0 = lo_bf(x2[k]), b1 = hi_bf(x2[k]);
                const float c0 = c[2 * k], c1 = c[2 * k + 1], s0 = sv[2 * k], s1 = sv[2 * k + 1];
                // low half: q*cos + (-q_hi)*sin ; high half: q_hi*cos + q_lo*sin
                o1[k] = pack2bf(bfround(a0 * c0) + bfround(-b0 * s0), bfround(a1 * c1) + bfround(-b1 * s1));
                o2[k] = pack2bf(bfround(b0 * c0) + bfround(a0 * s0), bfround(b1 * c1) + bfround(a1 * s1));
            }
            x1 = o1; x2 = o2;
            *(u32x4*)base = x1;
            *(u32x4*)(base + half) = x2;
        }
        if (hh >= nq && kcache != nullptr && p >= 0 && p < max_ctx) {
            const bool isv = hh >= nq + nkv;
            const int kvh = isv ? hh - nq - nkv : hh - nq;
            bf16_t* dst = (isv ? vcache : kcache) + (((int64_t)sq * nkv + kvh) * max_ctx + p) * hd + ch * 8;
            *(u32x4*)dst = x1;
            *(u32x4*)(dst + half) = x2;
        }
    }
}
int launch_rope_kv(bf16_t* qkv, const float* cs, const float* sn, const int32_t* pos, const int32_t* seq_of_tok,
                   bf16_t* kcache, bf16_t* vcache, int S, int nq, int nkv, int hd, int max_ctx, hipStream_t s) {
    VILA_REQUIRE(hd % 16 == 0, "rope: head_dim %d must be a multiple of 16", hd);
    const int64_t total = (int64_t)S * (nq + 2 * nkv) * (hd / 16);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(rope_kv_kernel, dim3(grid), dim3(256), 0, s, qkv, cs, sn, pos, seq_of_tok, kcache, vcache, S, nq, nkv, hd, max_ctx);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// row gathers: embedding lookup (llava_arch.py:429) and the media/text splice of _embed (llava_arch.py:457-479):
// dst[dst_row[i]] = src[src_row[i]]  (src_row null => i ; dst_row null => i)
// ------------------------------------------------------------------------------------------------
__global__ void copy_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const int32_t* __restrict__ src_row,
                                 const int32_t* __restrict__ dst_row, int n, int H, int64_t src_rows_max) {
    const int c8 = H >> 3;
    const int64_t total = (int64_t)n * c8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / c8), c = (int)(i % c8);
        int64_t sr = src_row != nullptr ? src_row[r] : r;
        if (sr < 0) sr = 0;
        if (src_rows_max > 0 && sr >= src_rows_max) sr = src_rows_max - 1;
        const int64_t dr = dst_row != nullptr ? dst_row[r] : r;
        *(u32x4*)(dst + dr * H + c * 8) = *(const u32x4*)(src + sr * H + c * 8);
    }
}
int launch_copy_rows(const bf16_t* src, bf16_t* dst, const int32_t* src_row, const int32_t* dst_row, int n, int H, hipStream_t s) {
    if (n == 0) return 0;
    VILA_REQUIRE(H % 8 == 0, "copy_rows: H=%d must be a multiple of 8", H);
    const int64_t total = (int64_t)n * (H / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid), dim3(256), 0, s, src, dst, src_row, dst_row, n, H, (int64_t)0);
    VILA_LAUNCH_CHECK();
    return 0;
}
__global__ void embed_gather_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ ids, bf16_t* __restrict__ out, int n, int H, int64_t vocab) {
    const int c8 = H >> 3;
    const int64_t total = (int64_t)n * c8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / c8), c = (int)(i % c8);
        int64_t id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        *(u32x4*)(out + (int64_t)r * H + c * 8) = *(const u32x4*)(table + id * H + c * 8);
    }
}
int launch_embed_gather(const bf16_t* table, const int64_t* ids, bf16_t* out, int n, int H, int64_t vocab, hipStream_t s) {
    if (n == 0) return 0;
    VILA_REQUIRE(H % 8 == 0, "embed: H=%d must be a multiple of 8", H);
    const int64_t total = (int64_t)n * (H / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(embed_gather_kernel, dim3(grid), dim3(256), 0, s, table, ids, out, n, H, vocab);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// greedy argmax over fp32 logits (HF GenerationMixin greedy, do_sample=False): first index of the maximum.
// ------------------------------------------------------------------------------------------------
#define ARGMAX_BLOCKS 256
__device__ __forceinline__ void amax_merge(float& v, int& i, float v2, int i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}
__global__ __launch_bounds__(256) void argmax_stage1(const float* __restrict__ logits, int V, float* __restrict__ tv, int* __restrict__ ti) {
    __shared__ float sv[4];
    __shared__ int si[4];
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < V; i += ARGMAX_BLOCKS * 256) amax_merge(best, bi, logits[i], i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64); const int i2 = __shfl_xor(bi, o, 64);
        amax_merge(best, bi, v2, i2);
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) amax_merge(best, bi, sv[w], si[w]);
        tv[blockIdx.x] = best; ti[blockIdx.x] = bi;
    }
}
__global__ __launch_bounds__(256) void argmax_stage2(const float* __restrict__ tv, const int* __restrict__ ti, int64_t* __restrict__ out) {
    __shared__ float sv[4];
    __shared__ int si[4];
    float best = tv[threadIdx.x]; int bi = ti[threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64); const int i2 = __shfl_xor(bi, o, 64);
        amax_merge(best, bi, v2, i2);
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) amax_merge(best, bi, sv[w], si[w]);
        out[0] = (int64_t)bi;
    }
}
int launch_argmax(const float* logits, int V, int64_t* out, float* tmpv, int* tmpi, hipStream_t s) {
    hipLaunchKernelGGL(argmax_stage1, dim3(ARGMAX_BLOCKS), dim3(256), 0, s, logits, V, tmpv, tmpi);
    VILA_LAUNCH_CHECK();
    hipLaunchKernelGGL(argmax_stage2, dim3(1), dim3(256), 0, s, tmpv, tmpi, out);
    VILA_LAUNCH_CHECK();
    return 0;
}
