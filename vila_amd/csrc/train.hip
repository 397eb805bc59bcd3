// Backward / optimizer kernels of the SFT step.  The dense contractions of backward reuse
// gemm_bf16_tn: dgrad  dX[M,K] = dY[M,N] . (W^T)[K,N]^T  and  wgrad  dW[N,K] = (dY^T)[N,M] . (X^T)[K,M]^T,  with the
// transposed operands produced by transpose_kernel below (activations are a few MB; the weight transposes cost
// ~2 x 16 GB of HBM traffic per step = ~5 ms, 2 % of the step).  Everything elementwise / reductions lives here.
#include "kernels.h"
#include "train.h"

// ------------------------------------------------------------------------------------------------
// bf16 2-D transpose through a padded LDS tile (64x64), 16-B global accesses on both sides when R,C % 8 == 0
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R, int C,
                                                        int64_t ldi, int64_t ldo) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    // load: 64 rows x 8 chunks of 8
    for (int i = tid; i < 512; i += 256) {
        const int r = i >> 3, ch = i & 7;
        const int gr = r0 + r, gc = c0 + ch * 8;
        u32x4 v = (u32x4){0u, 0u, 0u, 0u};
        if (gr < R && gc < C) v = *(const u32x4*)(in + (int64_t)gr * ldi + gc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            tile[r][ch * 8 + 2 * k] = (bf16_t)(v[k] & 0xffffu);
            tile[r][ch * 8 + 2 * k + 1] = (bf16_t)(v[k] >> 16);
        }
    }
    __syncthreads();
    const int Rp = (int)ldo;                         // output columns [R, ldo) are zero-filled (padding of the wgrad contraction dim)
    for (int i = tid; i < 512; i += 256) {
        const int c = i >> 3, ch = i & 7;          // output row = input column
        const int gc = c0 + c, gr = r0 + ch * 8;
        if (gc < C && gr < Rp) {
            u32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = (uint32_t)tile[ch * 8 + 2 * k][c] | ((uint32_t)tile[ch * 8 + 2 * k + 1][c] << 16);
            *(u32x4*)(out + (int64_t)gc * ldo + gr) = v;
        }
    }
}
int launch_transpose(const bf16_t* in, bf16_t* out, int R, int C, int64_t ldi, int64_t ldo, hipStream_t s) {
    VILA_REQUIRE(C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0 && ldo >= ((R + 7) & ~7), "transpose: C (%d), ld must be multiples of 8 and ldo >= round_up(R,8)", C);
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 64), cdiv((int)ldo, 64)), dim3(256), 0, s, in, out, R, C, ldi, ldo);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// elementwise helpers (vectorised 16 B / lane, grid-stride)
// ------------------------------------------------------------------------------------------------
#define EW_GRID(total) ((int)(((total) + 255) / 256 < 8192 ? ((total) + 255) / 256 : 8192))
#define EW_LOOP(total) for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

__device__ __forceinline__ float dgelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float t = 2.f * __builtin_amdgcn_rcpf(1.f + __expf(-2.f * k0 * (x + k1 * x * x * x))) - 1.f;      // tanh(u) = 2 sigmoid(2u) - 1 (common.h gelu_tanh_f)
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * k0 * (1.f + 3.f * k1 * x * x);
}
__device__ __forceinline__ float dgelu_erf(float x) {
    return 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// y = act(z)  (forward of the un-fused training path; z kept for backward)
template <int ACT>
__global__ void act_fwd_kernel(const bf16_t* __restrict__ z, bf16_t* __restrict__ y, int64_t n8) {
    EW_LOOP(n8) {
        const u32x4 v = *(const u32x4*)(z + i * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = lo_bf(v[k]), b = hi_bf(v[k]);
            o[k] = ACT == 1 ? pack2bf(gelu_tanh_f(a), gelu_tanh_f(b)) : pack2bf(gelu_erf_f(a), gelu_erf_f(b));
        }
        *(u32x4*)(y + i * 8) = o;
    }
}
// dz = dy * act'(z)
template <int ACT>
__global__ void act_bwd_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dz, int64_t n8) {
    EW_LOOP(n8) {
        const u32x4 v = *(const u32x4*)(z + i * 8);
        const u32x4 g = *(const u32x4*)(dy + i * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = lo_bf(v[k]), b = hi_bf(v[k]);
            const float da = ACT == 1 ? dgelu_tanh(a) : dgelu_erf(a), db = ACT == 1 ? dgelu_tanh(b) : dgelu_erf(b);
            o[k] = pack2bf(lo_bf(g[k]) * da, hi_bf(g[k]) * db);
        }
        *(u32x4*)(dz + i * 8) = o;
    }
}
int launch_act_fwd(const bf16_t* z, bf16_t* y, int64_t n, int act, hipStream_t s) {
    VILA_REQUIRE(n % 8 == 0 && (act == 1 || act == 2), "act_fwd: n %% 8 and act in {1,2}");
    if (act == 1) hipLaunchKernelGGL(act_fwd_kernel<1>, dim3(EW_GRID(n / 8)), dim3(256), 0, s, z, y, n / 8);
    else hipLaunchKernelGGL(act_fwd_kernel<2>, dim3(EW_GRID(n / 8)), dim3(256), 0, s, z, y, n / 8);
    VILA_LAUNCH_CHECK();
    return 0;
}
int launch_act_bwd(const bf16_t* z, const bf16_t* dy, bf16_t* dz, int64_t n, int act, hipStream_t s) {
    VILA_REQUIRE(n % 8 == 0 && (act == 1 || act == 2), "act_bwd: n %% 8 and act in {1,2}");
    if (act == 1) hipLaunchKernelGGL(act_bwd_kernel<1>, dim3(EW_GRID(n / 8)), dim3(256), 0, s, z, dy, dz, n / 8);
    else hipLaunchKernelGGL(act_bwd_kernel<2>, dim3(EW_GRID(n / 8)), dim3(256), 0, s, z, dy, dz, n / 8);
    VILA_LAUNCH_CHECK();
    return 0;
}

// SwiGLU: a = bf16(silu(g)) * u (HF rounding);  backward: dg = da * u * silu'(g), du = da * silu(g)
__global__ void silu_mul_fwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ u, bf16_t* __restrict__ a, int64_t n8) {
    EW_LOOP(n8) {
        const u32x4 gv = *(const u32x4*)(g + i * 8), uv = *(const u32x4*)(u + i * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack2bf(bfround(silu_f(lo_bf(gv[k]))) * lo_bf(uv[k]), bfround(silu_f(hi_bf(gv[k]))) * hi_bf(uv[k]));
        *(u32x4*)(a + i * 8) = o;
    }
}
__device__ __forceinline__ float dsilu(float x) { const float s = 1.f / (1.f + __expf(-x)); return s * (1.f + x * (1.f - s)); }
__global__ void silu_mul_bwd_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ u, const bf16_t* __restrict__ da,
                                    bf16_t* __restrict__ dg, bf16_t* __restrict__ du, int64_t n8) {
    EW_LOOP(n8) {
        const u32x4 gv = *(const u32x4*)(g + i * 8), uv = *(const u32x4*)(u + i * 8), dv = *(const u32x4*)(da + i * 8);
        u32x4 og, ou;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g0 = lo_bf(gv[k]), g1 = hi_bf(gv[k]), d0 = lo_bf(dv[k]), d1 = hi_bf(dv[k]);
            og[k] = pack2bf(d0 * lo_bf(uv[k]) * dsilu(g0), d1 * hi_bf(uv[k]) * dsilu(g1));
            ou[k] = pack2bf(d0 * silu_f(g0), d1 * silu_f(g1));
        }
        *(u32x4*)(dg + i * 8) = og;
        *(u32x4*)(du + i * 8) = ou;
    }
}
int launch_silu_mul_fwd(const bf16_t* g, const bf16_t* u, bf16_t* a, int64_t n, hipStream_t s) {
    VILA_REQUIRE(n % 8 == 0, "silu_mul: n %% 8");
    hipLaunchKernelGGL(silu_mul_fwd_kernel, dim3(EW_GRID(n / 8)), dim3(256), 0, s, g, u, a, n / 8);
    VILA_LAUNCH_CHECK();
    return 0;
}
int launch_silu_mul_bwd(const bf16_t* g, const bf16_t* u, const bf16_t* da, bf16_t* dg, bf16_t* du, int64_t n, hipStream_t s) {
    VILA_REQUIRE(n % 8 == 0, "silu_mul_bwd: n %% 8");
    hipLaunchKernelGGL(silu_mul_bwd_kernel, dim3(EW_GRID(n / 8)), dim3(256), 0, s, g, u, da, dg, du, n / 8);
    VILA_LAUNCH_CHECK();
    return 0;
}

// y = a + b (bf16)
__global__ void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int64_t n8) {
    EW_LOOP(n8) {
        const u32x4 av = *(const u32x4*)(a + i * 8), bv = *(const u32x4*)(b + i * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack2bf(lo_bf(av[k]) + lo_bf(bv[k]), hi_bf(av[k]) + hi_bf(bv[k]));
        *(u32x4*)(y + i * 8) = o;
    }
}
int launch_add(const bf16_t* a, const bf16_t* b, bf16_t* y, int64_t n, hipStream_t s) {
    VILA_REQUIRE(n % 8 == 0, "add: n %% 8");
    hipLaunchKernelGGL(add_kernel, dim3(EW_GRID(n / 8)), dim3(256), 0, s, a, b, y, n / 8);
    VILA_LAUNCH_CHECK();
    return 0;
}

// gradient accumulation over the micro-batches of one update (SFTTrainer.step_accumulated): the running sum is fp32, so 8-16 micro-batches
// (the NVILA scripts' --gradient_accumulation_steps) round to bf16 ONCE, when the last micro-batch's bucket is handed to the exchange.
//   MODE 0: acc = g            MODE 1: acc += g            MODE 2: out = bf16(acc + g)   (acc untouched; out may alias g)
template <int MODE>
__global__ void grad_accum_kernel(float* __restrict__ acc, const bf16_t* g, bf16_t* out, int64_t n8) {
    EW_LOOP(n8) {
        const u32x4 gv = *(const u32x4*)(g + i * 8);
        f32x4 a0, a1;
        if (MODE != 0) { a0 = *(const f32x4*)(acc + i * 8); a1 = *(const f32x4*)(acc + i * 8 + 4); }
        else { a0 = f32x4{0.f, 0.f, 0.f, 0.f}; a1 = a0; }
        a0[0] += lo_bf(gv[0]); a0[1] += hi_bf(gv[0]); a0[2] += lo_bf(gv[1]); a0[3] += hi_bf(gv[1]);
        a1[0] += lo_bf(gv[2]); a1[1] += hi_bf(gv[2]); a1[2] += lo_bf(gv[3]); a1[3] += hi_bf(gv[3]);
        if (MODE == 2) {
            u32x4 o;
            o[0] = pack2bf(a0[0], a0[1]); o[1] = pack2bf(a0[2], a0[3]); o[2] = pack2bf(a1[0], a1[1]); o[3] = pack2bf(a1[2], a1[3]);
            *(u32x4*)(out + i * 8) = o;
        } else {
            *(f32x4*)(acc + i * 8) = a0; *(f32x4*)(acc + i * 8 + 4) = a1;
        }
    }
}
int launch_grad_accum(float* acc, const bf16_t* g, bf16_t* out, int64_t n, int mode, hipStream_t s) {
    VILA_REQUIRE(n % 8 == 0, "grad_accum: n %% 8");
    VILA_REQUIRE(mode >= 0 && mode <= 2 && (mode != 2 || out != nullptr), "grad_accum: mode 0 / 1 / 2 (2 needs out)");
    if (n == 0) return 0;
    const dim3 g_(EW_GRID(n / 8)), b_(256);
    if (mode == 0) hipLaunchKernelGGL(grad_accum_kernel<0>, g_, b_, 0, s, acc, g, out, n / 8);
    else if (mode == 1) hipLaunchKernelGGL(grad_accum_kernel<1>, g_, b_, 0, s, acc, g, out, n / 8);
    else hipLaunchKernelGGL(grad_accum_kernel<2>, g_, b_, 0, s, acc, g, out, n / 8);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// column sums: out[c] (+)= sum_r x[r][c]  (bias / position-embedding gradients).  One block per 64 columns; each of the
// 4 waves walks a quarter of the rows with lane = column; fp32 accumulate.
// ------------------------------------------------------------------------------------------------
// stage 1: block = 128 columns x (R / gridDim.y) rows; thread = (row lane, 8-column chunk), 16-B loads, fp32 partials reduced through LDS,
// ONE fp32 partial per column and block into part[blockIdx.y][C]; stage 2 (colpart_reduce_kernel) adds the <= 64 partials of a column in a
// FIXED order and converts.  (Rounds 2-5 used one fp32 atomicAdd per column and block: sums depended on arrival order, so two identical SFT
// steps differed in the last bits and a resumed run could not be compared bit for bit with the uninterrupted one — VERDICT round 5.)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, int R, int C, int64_t ld) {
    __shared__ float sm[16][132];
    const int tid = threadIdx.x, cg = tid & 15, rl = tid >> 4;
    const int c0 = blockIdx.x * 128 + cg * 8;
    const int rows_per = (R + gridDim.y - 1) / gridDim.y;
    const int rb = blockIdx.y * rows_per, re = (rb + rows_per < R) ? rb + rows_per : R;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    if (c0 < C) {
        for (int r = rb + rl; r < re; r += 16) {
            const u32x4 v = *(const u32x4*)(x + (int64_t)r * ld + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[2 * k] += lo_bf(v[k]); acc[2 * k + 1] += hi_bf(v[k]); }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[rl][cg * 8 + k] = acc[k];
    __syncthreads();
    if (tid < 128) {
        const int c = blockIdx.x * 128 + tid;
        if (c < C) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v += sm[r][tid];
            part[(int64_t)blockIdx.y * C + c] = v;
        }
    }
}
// out[c] (+)= sum_p part[p][c], p ascending inside each of 4 interleaved groups, then the 4 group sums in order: a fixed summation tree.
// block = 64 columns x 4 partial groups
__global__ __launch_bounds__(256) void colpart_reduce_kernel(const float* __restrict__ part, int nparts, int C, bf16_t* __restrict__ out, int accumulate) {
    __shared__ float sm[4][64];
    const int tid = threadIdx.x, cl = tid & 63, grp = tid >> 6;
    const int c = blockIdx.x * 64 + cl;
    float v = 0.f;
    if (c < C)
        for (int p = grp; p < nparts; p += 4) v += part[(int64_t)p * C + c];
    sm[grp][cl] = v;
    __syncthreads();
    if (grp == 0 && c < C) {
        const float t = (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
        out[c] = f2bf(t + (accumulate ? bf2f(out[c]) : 0.f));
    }
}
static int launch_colpart_reduce(const float* part, int nparts, int C, bf16_t* out, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(colpart_reduce_kernel, dim3(cdiv(C, 64)), dim3(256), 0, s, part, nparts, C, out, accumulate);
    VILA_LAUNCH_CHECK();
    return 0;
}
static inline int colsum_parts(int R) { int gy = cdiv(R, 128); return gy > 64 ? 64 : (gy < 1 ? 1 : gy); }
size_t colsum_scratch_floats(int R, int C) { return (size_t)colsum_parts(R) * C; }
// out[p][c] (+)= sum_b x[b*period + p][c]   (position-embedding gradient: sum over images)
__global__ void periodic_sum_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int nrep, int64_t n8, int accumulate) {
    EW_LOOP(n8) {
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int b = 0; b < nrep; ++b) {
            const u32x4 v = *(const u32x4*)(x + ((int64_t)b * n8 + i) * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[2 * k] += lo_bf(v[k]); acc[2 * k + 1] += hi_bf(v[k]); }
        }
        u32x4 o;
        if (accumulate) {
            const u32x4 p = *(const u32x4*)(out + i * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { acc[2 * k] += lo_bf(p[k]); acc[2 * k + 1] += hi_bf(p[k]); }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = pack2bf(acc[2 * k], acc[2 * k + 1]);
        *(u32x4*)(out + i * 8) = o;
    }
}
__global__ void f32_to_bf16_acc_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int n, int accumulate);
int launch_colsum(const bf16_t* x, bf16_t* out, float* scratch, int R, int C, int64_t ld, int accumulate, int period, hipStream_t s) {
    VILA_REQUIRE(C % 8 == 0 && ld % 8 == 0, "colsum: C (%d) and ld must be multiples of 8", C);
    if (period > 0) {
        VILA_REQUIRE(R % period == 0 && ld == C, "colsum: periodic mode needs R %% period == 0 and a dense input");
        const int64_t n8 = (int64_t)period * C / 8;
        hipLaunchKernelGGL(periodic_sum_kernel, dim3(EW_GRID(n8)), dim3(256), 0, s, x, out, R / period, n8, accumulate);
        VILA_LAUNCH_CHECK();
        return 0;
    }
    VILA_REQUIRE(scratch != nullptr, "colsum: fp32 scratch of vila_colsum_scratch_floats(rows, cols) floats required");
    const int gy = colsum_parts(R);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 128), gy), dim3(256), 0, s, x, scratch, R, C, ld);
    VILA_LAUNCH_CHECK();
    return launch_colpart_reduce(scratch, gy, C, out, accumulate, s);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm backward.  dx per row; every block writes ONE fp32 partial of dw (db) per column into part[block][cols] and
// colpart_reduce_kernel adds the partials of a column in a fixed order (deterministic: no atomics — see colsum above).
//   LN : xhat = (x-mean)*rstd ; dx = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*w ; dw += dy*xhat ; db += dy
//   RMS: xhat = x*rstd        ; dx = rstd*(g - xhat*mean(g*xhat))             ; dw += dy*xhat
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum4(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
// generic kernel: block (x = 8 rows, y = chunk of 4096 columns): the row statistics are formed over the WHOLE row by every column chunk's block
// (rows wider than 4096: the projector's 4608 / 13 824-wide LayerNorm), dx / dw / db only for the block's own columns
template <bool RMS>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ dy,
                                                       bf16_t* __restrict__ dx, float* __restrict__ dw_part, float* __restrict__ db_part,
                                                       int rows, int cols, float eps) {
    __shared__ float scratch[4];
    const int tid = threadIdx.x;
    constexpr int NORM_ROWS = 8, NJ = 16;              // 16 column slots per thread = 4096 columns per block
    const int cb = blockIdx.y * 256 * NJ, ce = (cb + 256 * NJ < cols) ? cb + 256 * NJ : cols;
    float dwacc[NJ], dbacc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { dwacc[j] = 0.f; dbacc[j] = 0.f; }
  for (int rowi = 0; rowi < NORM_ROWS; ++rowi) {
    const int row = blockIdx.x * NORM_ROWS + rowi;
    if (row >= rows) break;
    const bf16_t* xr = x + (int64_t)row * cols;
    const bf16_t* gr = dy + (int64_t)row * cols;
    bf16_t* dr = dx + (int64_t)row * cols;
    float s1 = 0.f, s2 = 0.f;
    for (int c = tid; c < cols; c += 256) { const float v = bf2f(xr[c]); s1 += v; s2 += v * v; }
    s1 = block_sum4(s1, scratch);
    s2 = block_sum4(s2, scratch);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s2 / cols + eps);
    } else {
        mean = s1 / cols;
        float q = 0.f;
        for (int c = tid; c < cols; c += 256) { const float d = bf2f(xr[c]) - mean; q += d * d; }
        q = block_sum4(q, scratch);
        rstd = rsqrtf(q / cols + eps);
    }
    float a = 0.f, b = 0.f;     // a = sum(g), b = sum(g*xhat)
    for (int c = tid; c < cols; c += 256) {
        const float xh = (bf2f(xr[c]) - mean) * rstd;
        const float g = bf2f(gr[c]) * bf2f(w[c]);
        a += g; b += g * xh;
    }
    a = block_sum4(a, scratch);
    b = block_sum4(b, scratch);
    const float ma = RMS ? 0.f : a / cols, mb = b / cols;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = cb + tid + 256 * j;
        if (c < ce) {
            const float xh = (bf2f(xr[c]) - mean) * rstd;
            const float dyv = bf2f(gr[c]);
            const float g = dyv * bf2f(w[c]);
            dr[c] = f2bf(rstd * (g - ma - xh * mb));
            dwacc[j] += dyv * xh; dbacc[j] += dyv;
        }
    }
    __syncthreads();
  }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = cb + tid + 256 * j;
        if (c < ce) {
            dw_part[(int64_t)blockIdx.x * cols + c] = dwacc[j];
            if (!RMS && db_part != nullptr) db_part[(int64_t)blockIdx.x * cols + c] = dbacc[j];
        }
    }
}
__global__ void f32_to_bf16_acc_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int n, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f2bf(src[i] + (accumulate ? bf2f(dst[i]) : 0.f));
}
// Wave-per-row variant for cols <= 512 * MAXI (the RMSNorm / LayerNorm widths of the model: 3584, 1152): a wave keeps the row's x
// and dy in registers (16-B loads, one pass over HBM), all row reductions are wave-level (no block barriers), every wave walks
// NB_ROWS rows accumulating its dw / db columns in registers; the four waves of a block meet once in LDS and the block writes
// one fp32 partial per column (part[block][cols]).  (The generic kernel above re-reads the row 4x with 2-B loads and crosses ~10 block barriers per
// row: 127 us for [3076, 3584] vs ~20 us here.)
#define NB_ROWS 3
template <bool RMS, int MAXI>
__global__ __launch_bounds__(256) void norm_bwd_wave_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const bf16_t* __restrict__ dy,
                                                            bf16_t* __restrict__ dx, float* __restrict__ dw32, float* __restrict__ db32,
                                                            int rows, int cols, float eps) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4][cols] (+ [4][cols] for db)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = cols >> 3;
    const float inv_cols = 1.f / (float)cols;
    u32x4 wv[MAXI];
    float dwacc[MAXI][8], dbacc[RMS ? 1 : MAXI][8];
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = lane + 64 * i;
        wv[i] = (c < nch) ? *(const u32x4*)(w + c * 8) : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) { dwacc[i][e] = 0.f; if (!RMS) dbacc[i][e] = 0.f; }
    }
    const int row0 = (blockIdx.x * 4 + wave) * NB_ROWS;
#pragma unroll 1
    for (int r = 0; r < NB_ROWS; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;                                     // wave-uniform
        const bf16_t* xr = x + (int64_t)row * cols;
        const bf16_t* gr = dy + (int64_t)row * cols;
        u32x4 xv[MAXI], gv[MAXI];
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c = lane + 64 * i;
            xv[i] = (c < nch) ? *(const u32x4*)(xr + c * 8) : (u32x4){0u, 0u, 0u, 0u};
            gv[i] = (c < nch) ? *(const u32x4*)(gr + c * 8) : (u32x4){0u, 0u, 0u, 0u};
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float a = lo_bf(xv[i][k]), b = hi_bf(xv[i][k]); s1 += a + b; s2 += a * a + b * b; }
        float mean = 0.f, rstd;
        if (RMS) {
            rstd = rsqrtf(wave_sum(s2) * inv_cols + eps);
        } else {
            mean = wave_sum(s1) * inv_cols;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                if (lane + 64 * i < nch) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const float a = lo_bf(xv[i][k]) - mean, b = hi_bf(xv[i][k]) - mean; q += a * a + b * b; }
                }
            }
            rstd = rsqrtf(wave_sum(q) * inv_cols + eps);
        }
        float sa = 0.f, sb = 0.f;                                   // sum(g), sum(g * xhat), g = dy * w
#pragma unroll
        for (int i = 0; i < MAXI; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g0 = lo_bf(gv[i][k]) * lo_bf(wv[i][k]), g1 = hi_bf(gv[i][k]) * hi_bf(wv[i][k]);
                const float x0 = (lo_bf(xv[i][k]) - mean) * rstd, x1 = (hi_bf(xv[i][k]) - mean) * rstd;
                if (lane + 64 * i < nch) { sa += g0 + g1; sb += g0 * x0 + g1 * x1; }
            }
        const float ma = RMS ? 0.f : wave_sum(sa) * inv_cols, mb = wave_sum(sb) * inv_cols;
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d0 = lo_bf(gv[i][k]), d1 = hi_bf(gv[i][k]);
                    const float x0 = (lo_bf(xv[i][k]) - mean) * rstd, x1 = (hi_bf(xv[i][k]) - mean) * rstd;
                    o[k] = pack2bf(rstd * (d0 * lo_bf(wv[i][k]) - ma - x0 * mb), rstd * (d1 * hi_bf(wv[i][k]) - ma - x1 * mb));
                    dwacc[i][2 * k] += d0 * x0; dwacc[i][2 * k + 1] += d1 * x1;
                    if (!RMS) { dbacc[i][2 * k] += d0; dbacc[i][2 * k + 1] += d1; }
                }
                *(u32x4*)(dx + (int64_t)row * cols + c * 8) = o;
            }
        }
    }
    // ---- four waves meet in LDS, one partial per column and block ----
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[wave * cols + c * 8 + e] = dwacc[i][e];
                if (!RMS) red[(4 + wave) * cols + c * 8 + e] = dbacc[i][e];
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < cols; c += 256) {
        dw32[(int64_t)blockIdx.x * cols + c] = (red[c] + red[cols + c]) + (red[2 * cols + c] + red[3 * cols + c]);
        if (!RMS && db32 != nullptr) db32[(int64_t)blockIdx.x * cols + c] = (red[4 * cols + c] + red[5 * cols + c]) + (red[6 * cols + c] + red[7 * cols + c]);
    }
}

template <bool RMS, int MAXI>
static void launch_norm_bwd_wave(const bf16_t* x, const bf16_t* w, const bf16_t* dy, bf16_t* dx, float* dw32, float* db32, int rows, int cols, float eps,
                                 hipStream_t s) {
    const size_t lds = (size_t)(RMS ? 4 : 8) * cols * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)norm_bwd_wave_kernel<RMS, MAXI>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024 * 2);
        attr_set = true;
    }
    hipLaunchKernelGGL((norm_bwd_wave_kernel<RMS, MAXI>), dim3(cdiv(rows, 4 * NB_ROWS)), dim3(256), lds, s, x, w, dy, dx, dw32, db32, rows, cols, eps);
}

// which kernel a shape takes and how many per-column partials it writes
static inline bool norm_bwd_wave_ok(int cols, int rms, bool aligned) { return aligned && ((rms && cols <= 4096) || (!rms && cols <= 1536)); }
static inline int norm_bwd_parts(int rows, int cols, int rms, bool aligned) { return norm_bwd_wave_ok(cols, rms, aligned) ? cdiv(rows, 4 * NB_ROWS) : cdiv(rows, 8); }
size_t norm_bwd_scratch_floats(int rows, int cols) {            // upper bound over both kernels (the generic one has the smaller row blocks), dw + db
    return (size_t)2 * cdiv(rows, 8) * cols;
}
int launch_norm_bwd(const bf16_t* x, const bf16_t* w, const bf16_t* dy, bf16_t* dx, bf16_t* dw, bf16_t* db, float* scratch /*norm_bwd_scratch_floats*/,
                    int rows, int cols, float eps, int rms, int accumulate, hipStream_t s) {
    const bool aligned = cols % 8 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dy % 16 == 0 && (uintptr_t)dx % 16 == 0 && (uintptr_t)w % 16 == 0;
    const int np = norm_bwd_parts(rows, cols, rms, aligned);
    float* dwp = scratch;
    float* dbp = scratch + (size_t)np * cols;
    if (aligned && rms && cols <= 3584) {
        launch_norm_bwd_wave<true, 7>(x, w, dy, dx, dwp, nullptr, rows, cols, eps, s);
    } else if (aligned && rms && cols <= 4096) {
        launch_norm_bwd_wave<true, 8>(x, w, dy, dx, dwp, nullptr, rows, cols, eps, s);
    } else if (aligned && !rms && cols <= 1536) {
        launch_norm_bwd_wave<false, 3>(x, w, dy, dx, dwp, dbp, rows, cols, eps, s);
    } else if (rms) {
        hipLaunchKernelGGL(norm_bwd_kernel<true>, dim3(np, cdiv(cols, 4096)), dim3(256), 0, s, x, w, dy, dx, dwp, (float*)nullptr, rows, cols, eps);
    } else {
        hipLaunchKernelGGL(norm_bwd_kernel<false>, dim3(np, cdiv(cols, 4096)), dim3(256), 0, s, x, w, dy, dx, dwp, dbp, rows, cols, eps);
    }
    VILA_LAUNCH_CHECK();
    VILA_TRY(launch_colpart_reduce(dwp, np, cols, dw, accumulate, s));
    if (!rms && db != nullptr) VILA_TRY(launch_colpart_reduce(dbp, np, cols, db, accumulate, s));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// softmax cross-entropy over fp32 logits rows (HF ForCausalLMLoss, reduction sum / num_items): in-place gradient.
//   loss += (lse - z[label]) * scale ;  dz = (softmax(z) - onehot(label)) * scale  written as bf16 into dlogits
// ------------------------------------------------------------------------------------------------
// The row's loss goes to row_loss[row]; ce_sum_kernel adds the rows in a fixed order (one block, fixed tree) into *loss.
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, bf16_t* __restrict__ dlogits,
                                                 float* __restrict__ row_loss, int V, int64_t ldl, float scale) {
    __shared__ float scratch[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* z = logits + (int64_t)row * ldl;
    bf16_t* dz = dlogits + (int64_t)row * V;
    const int64_t lab = labels[row];
    float m = -INFINITY;
    for (int c = tid; c < V; c += 256) m = fmaxf(m, z[c]);
    m = wave_max(m);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    float sum = 0.f;
    for (int c = tid; c < V; c += 256) sum += __expf(z[c] - m);
    sum = block_sum4(sum, scratch);
    const float inv = 1.f / sum;
    const bool valid = lab >= 0 && lab < V;
    for (int c = tid; c < V; c += 256) {
        float p = __expf(z[c] - m) * inv;
        if (c == lab) p -= 1.f;
        dz[c] = f2bf(valid ? p * scale : 0.f);
    }
    if (tid == 0) row_loss[row] = valid ? (m + logf(sum) - z[lab]) * scale : 0.f;
}
// *acc += sum_i v[i]: thread t adds v[t], v[t + 256], ... in order, then the block's fixed tree (also the second stage of sumsq)
__global__ __launch_bounds__(256) void ordered_sum_kernel(const float* __restrict__ v, int n, float* __restrict__ acc) {
    __shared__ float scratch[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += v[i];
    a = block_sum4(a, scratch);
    if (threadIdx.x == 0) *acc += a;
}
int launch_ce(const float* logits, const int64_t* labels, bf16_t* dlogits, float* loss, float* row_loss, int rows, int V, int64_t ldl, float scale, hipStream_t s) {
    if (rows == 0) return 0;
    VILA_REQUIRE(row_loss != nullptr, "ce_loss: a scratch of `rows` floats is required (the per-row losses, summed in a fixed order)");
    hipLaunchKernelGGL(ce_kernel, dim3(rows), dim3(256), 0, s, logits, labels, dlogits, row_loss, V, ldl, scale);
    VILA_LAUNCH_CHECK();
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, s, row_loss, rows, loss);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// scatter-add rows into a bf16 table (embedding gradient): dst[rows[i]] += src[i].  Deterministic and single-writer: block i owns destination
// row rows[i] iff no j < i has the same id; the owner adds the source rows of ALL occurrences j >= i in ascending j order in fp32 and rounds once
// (rounds 2-5: a packed-bf16 CAS loop — duplicates, i.e. every repeated token of a batch, were added in arrival order and rounded per add).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, const int32_t* __restrict__ rows, int n, int H) {
    __shared__ int flags[2];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int id = rows[i];
    if (tid < 2) flags[tid] = 0;
    __syncthreads();
    int earlier = 0, later = 0;
    for (int j = tid; j < n; j += 256) {
        if (rows[j] == id) { if (j < i) earlier = 1; else if (j > i) later = 1; }
    }
    if (earlier) flags[0] = 1;
    if (later) flags[1] = 1;
    __syncthreads();
    if (flags[0]) return;                                        // an earlier occurrence owns this destination row
    const bool dup = flags[1] != 0;
    const int w2 = H >> 1;
    for (int c = tid; c < w2; c += 256) {
        const uint32_t own = *(const uint32_t*)(src + (int64_t)i * H + 2 * c);
        float a0 = lo_bf(own), a1 = hi_bf(own);
        if (dup) {
            for (int j = i + 1; j < n; ++j) {
                if (rows[j] == id) {                             // block-uniform
                    const uint32_t v = *(const uint32_t*)(src + (int64_t)j * H + 2 * c);
                    a0 += lo_bf(v); a1 += hi_bf(v);
                }
            }
        }
        uint32_t* d = (uint32_t*)(dst + (int64_t)id * H + 2 * c);
        const uint32_t old = *d;
        *d = pack2bf(lo_bf(old) + a0, hi_bf(old) + a1);
    }
}
int launch_scatter_add_rows(const bf16_t* src, bf16_t* dst, const int32_t* rows, int n, int H, hipStream_t s) {
    if (n == 0) return 0;
    VILA_REQUIRE(H % 2 == 0, "scatter_add_rows: H must be even");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(n), dim3(256), 0, s, src, dst, rows, n, H);
    VILA_LAUNCH_CHECK();
    return 0;
}

// depth-to-space = backward of flat_square k x k (padded cells receive no gradient)
__global__ void d2s_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int B, int g, int C, int k) {
    const int gd = (g + k - 1) / k, c8 = C >> 3;
    const int64_t total = (int64_t)B * g * g * c8;
    EW_LOOP(total) {
        int64_t r = i;
        const int ch = (int)(r % c8); r /= c8;
        const int sx = (int)(r % g); r /= g;
        const int sy = (int)(r % g); r /= g;
        const int b = (int)r;
        const int ii = sy / k, a = sy % k, j = sx / k, bb = sx % k;
        *(u32x4*)(dx + ((int64_t)b * g * g + sy * g + sx) * C + ch * 8) =
            *(const u32x4*)(dy + (((int64_t)b * gd * gd + ii * gd + j) * k * k + a * k + bb) * C + ch * 8);
    }
}
int launch_depth_to_space(const bf16_t* dy, bf16_t* dx, int B, int g, int C, int k, hipStream_t s) {
    VILA_REQUIRE(C % 8 == 0, "depth_to_space: C %% 8");
    hipLaunchKernelGGL(d2s_kernel, dim3(EW_GRID((int64_t)B * g * g * C / 8)), dim3(256), 0, s, dy, dx, B, g, C, k);
    VILA_LAUNCH_CHECK();
    return 0;
}

// RoPE backward on a fused dqkv buffer [S][q+2kv]: the rotation matrix transposed = same rotation with -sin (q and k heads)
__global__ void rope_bwd_kernel(bf16_t* __restrict__ dqkv, const float* __restrict__ cs, const float* __restrict__ sn, int S, int nq, int nkv, int hd) {
    const int half = hd >> 1, cpr = half >> 3, heads = nq + nkv;
    const int64_t total = (int64_t)S * heads * cpr;
    const int row = (nq + 2 * nkv) * hd;
    EW_LOOP(total) {
        int64_t r = i;
        const int ch = (int)(r % cpr); r /= cpr;
        const int hh = (int)(r % heads); r /= heads;
        const int s = (int)r;
        bf16_t* base = dqkv + (int64_t)s * row + hh * hd + ch * 8;
        const u32x4 x1 = *(const u32x4*)base, x2 = *(const u32x4*)(base + half);
        const float* c = cs + (int64_t)s * half + ch * 8;
        const float* sv = sn + (int64_t)s * half + ch * 8;
        u32x4 o1, o2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a0 = lo_bf(x1[k]), a1 = hi_bf(x1[k]), b0 = lo_bf(x2[k]), b1 = hi_bf(x2[k]);
            // forward: y1 = x1 c - x2 s ; y2 = x2 c + x1 s   =>   dx1 = dy1 c + dy2 s ; dx2 = dy2 c - dy1 s
            o1[k] = pack2bf(a0 * c[2 * k] + b0 * sv[2 * k], a1 * c[2 * k + 1] + b1 * sv[2 * k + 1]);
            o2[k] = pack2bf(b0 * c[2 * k] - a0 * sv[2 * k], b1 * c[2 * k + 1] - a1 * sv[2 * k + 1]);
        }
        *(u32x4*)base = o1;
        *(u32x4*)(base + half) = o2;
    }
}
int launch_rope_bwd(bf16_t* dqkv, const float* cs, const float* sn, int S, int nq, int nkv, int hd, hipStream_t s) {
    hipLaunchKernelGGL(rope_bwd_kernel, dim3(EW_GRID((int64_t)S * (nq + nkv) * hd / 16)), dim3(256), 0, s, dqkv, cs, sn, S, nq, nkv, hd);
    VILA_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW semantics, adamw_torch in llava/train/args.py:223): fp32 master weight + m + v, bf16 grad in,
// bf16 param out.  One flat launch over the whole model.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void adamw_one(float& p, float& mi, float& vi, float g, float lr, float b1, float b2, float eps, float wd, float bc1,
                                          float bc2) {
    mi = b1 * mi + (1.f - b1) * g;
    vi = b2 * vi + (1.f - b2) * g * g;
    p *= (1.f - lr * wd);
    p -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
}
// 4 parameters per lane: 16-B non-temporal accesses of master / m / v (read + write), 8 B of grad in and of the bf16 parameter out —
// 28 B per parameter, every byte touched once per step (nothing to keep in L2 / MALL)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const bf16_t* __restrict__ grad, bf16_t* __restrict__ param, int64_t n, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2, float grad_scale) {
    const int64_t n4 = n >> 2;
    EW_LOOP(n4) {
        f32x4 p = __builtin_nontemporal_load((const f32x4*)master + i);
        f32x4 mi = __builtin_nontemporal_load((const f32x4*)m + i);
        f32x4 vi = __builtin_nontemporal_load((const f32x4*)v + i);
        const u32x2 g2 = __builtin_nontemporal_load((const u32x2*)grad + i);
        const float g[4] = {lo_bf(g2[0]) * grad_scale, hi_bf(g2[0]) * grad_scale, lo_bf(g2[1]) * grad_scale, hi_bf(g2[1]) * grad_scale};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = p[k], mk = mi[k], vk = vi[k];
            adamw_one(pk, mk, vk, g[k], lr, b1, b2, eps, wd, bc1, bc2);
            p[k] = pk; mi[k] = mk; vi[k] = vk;
        }
        __builtin_nontemporal_store(mi, (f32x4*)m + i);
        __builtin_nontemporal_store(vi, (f32x4*)v + i);
        __builtin_nontemporal_store(p, (f32x4*)master + i);
        u32x2 o; o[0] = pack2bf(p[0], p[1]); o[1] = pack2bf(p[2], p[3]);
        __builtin_nontemporal_store(o, (u32x2*)param + i);
    }
    // tail (n % 4)
    const int64_t t = (n4 << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < n) {
        float p = master[t], mi = m[t], vi = v[t];
        adamw_one(p, mi, vi, bf2f(grad[t]) * grad_scale, lr, b1, b2, eps, wd, bc1, bc2);
        m[t] = mi; v[t] = vi; master[t] = p; param[t] = f2bf(p);
    }
}
int launch_adamw(float* master, float* m, float* v, const bf16_t* grad, bf16_t* param, int64_t n, float lr, float b1, float b2, float eps,
                 float wd, int step, float grad_scale, hipStream_t s) {
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    VILA_REQUIRE(((uintptr_t)master | (uintptr_t)m | (uintptr_t)v) % 16 == 0 && ((uintptr_t)grad | (uintptr_t)param) % 8 == 0, "adamw: buffers must be 16-B (fp32) / 8-B (bf16) aligned");
    hipLaunchKernelGGL(adamw_kernel, dim3(EW_GRID((n >> 2) > 0 ? (n >> 2) : 1)), dim3(256), 0, s, master, m, v, grad, param, n, lr, b1, b2, eps, wd, bc1, bc2, grad_scale);
    VILA_LAUNCH_CHECK();
    return 0;
}
// "Lean" AdamW for the optimizer stream of the SFT step: 2 parameters per lane, every array addressed through a buffer descriptor (base
// in SGPRs, ONE 32-bit VGPR offset) -> 29 VGPRs.  A 256x256 GEMM block keeps two ~240-VGPR waves on every SIMD (480 of the 512-entry
// register file); only a kernel that allocates <= 32 registers per lane can be co-resident with it, and then the optimizer's 28 B per
// parameter stream from HBM while the matrix cores work on the next layer.  (The 4-wide kernel above needs 55 VGPRs: it waits for the
// GEMM blocks to drain.)  nt cache policy (aux = 2) on every access, as above.
__global__ __launch_bounds__(256) void adamw_lean_kernel(float* master, float* m, float* v, const uint32_t* grad, uint32_t* param, unsigned n2,
                                                         float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float grad_scale,
                                                         int odd_tail) {
    if (odd_tail && blockIdx.x == 0 && threadIdx.x == 0) {           // element 2*n2 of an odd-length buffer
        const int64_t t = (int64_t)n2 * 2;
        float p = master[t], mi = m[t], vi = v[t];
        adamw_one(p, mi, vi, bf2f(((const bf16_t*)grad)[t]) * grad_scale, lr, b1, b2, eps, wd, bc1, bc2);
        m[t] = mi; v[t] = vi; master[t] = p; ((bf16_t*)param)[t] = f2bf(p);
    }
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(master, 0, n2 * 8u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(m, 0, n2 * 8u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(v, 0, n2 * 8u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)grad, 0, n2 * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(param, 0, n2 * 4u, 0x00020000);
    const unsigned stride = gridDim.x * blockDim.x;
#pragma clang loop unroll(disable)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const u32x2 pw = __builtin_amdgcn_raw_buffer_load_b64(rp, i * 8u, 0, 2);
        const u32x2 mw = __builtin_amdgcn_raw_buffer_load_b64(rm, i * 8u, 0, 2);
        const u32x2 vw = __builtin_amdgcn_raw_buffer_load_b64(rv, i * 8u, 0, 2);
        const unsigned g2 = __builtin_amdgcn_raw_buffer_load_b32(rg, i * 4u, 0, 2);
        float p0 = __uint_as_float(pw[0]), m0 = __uint_as_float(mw[0]), v0 = __uint_as_float(vw[0]);
        adamw_one(p0, m0, v0, lo_bf(g2) * grad_scale, lr, b1, b2, eps, wd, bc1, bc2);
        asm volatile("" : "+v"(p0), "+v"(m0), "+v"(v0));            // finish element 0 before element 1 starts: halves the live temporaries
        float p1 = __uint_as_float(pw[1]), m1 = __uint_as_float(mw[1]), v1 = __uint_as_float(vw[1]);
        adamw_one(p1, m1, v1, hi_bf(g2) * grad_scale, lr, b1, b2, eps, wd, bc1, bc2);
        const u32x2 po = {__float_as_uint(p0), __float_as_uint(p1)}, mo = {__float_as_uint(m0), __float_as_uint(m1)}, vo = {__float_as_uint(v0), __float_as_uint(v1)};
        __builtin_amdgcn_raw_buffer_store_b64(mo, rm, i * 8u, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b64(vo, rv, i * 8u, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b64(po, rp, i * 8u, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b32(pack2bf(p0, p1), rq, i * 4u, 0, 2);
    }
}
// Round 6: the form that is meant to run BESIDE a resident 256x256 GEMM block instead of between two of them.  A GEMM block holds 2 x 224-232
// VGPRs per SIMD lane and 128-132 KB of LDS, so exactly ONE block of this kernel fits next to it (<= 48 VGPRs, 28 KB of LDS requested and never
// touched: the request is what keeps a CU that is momentarily free from filling up with optimizer blocks a GEMM block would then have to wait
// out).  Blocks are SHORT (256 threads x U element pairs, all 7 U loads requested up front) — the persistent grid-stride kernel above keeps a CU's
// slots for ~0.4 ms per block and streams 1.8 KB per wave; this one streams 7 KB per wave and leaves after a few microseconds.
template <int U>
__global__ __launch_bounds__(256) void adamw_stream_kernel(float* master, float* m, float* v, const uint32_t* grad, uint32_t* param, unsigned n2,
                                                           float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float grad_scale) {
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(master, 0, n2 * 8u, 0x00020000);      // range-checked: loads beyond n2 read 0,
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(m, 0, n2 * 8u, 0x00020000);           // stores beyond n2 are dropped
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(v, 0, n2 * 8u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)grad, 0, n2 * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(param, 0, n2 * 4u, 0x00020000);
  for (unsigned base = blockIdx.x * (256u * U) + threadIdx.x; base - threadIdx.x < n2; base += gridDim.x * (256u * U)) {
    u32x2 pw[U], mw[U], vw[U];
    unsigned g2[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned i = base + 256u * u;
        pw[u] = __builtin_amdgcn_raw_buffer_load_b64(rp, i * 8u, 0, 2);
        mw[u] = __builtin_amdgcn_raw_buffer_load_b64(rm, i * 8u, 0, 2);
        vw[u] = __builtin_amdgcn_raw_buffer_load_b64(rv, i * 8u, 0, 2);
        g2[u] = __builtin_amdgcn_raw_buffer_load_b32(rg, i * 4u, 0, 2);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const unsigned i = base + 256u * u;
        float p0 = __uint_as_float(pw[u][0]), m0 = __uint_as_float(mw[u][0]), v0 = __uint_as_float(vw[u][0]);
        adamw_one(p0, m0, v0, lo_bf(g2[u]) * grad_scale, lr, b1, b2, eps, wd, bc1, bc2);
        asm volatile("" : "+v"(p0), "+v"(m0), "+v"(v0));
        float p1 = __uint_as_float(pw[u][1]), m1 = __uint_as_float(mw[u][1]), v1 = __uint_as_float(vw[u][1]);
        adamw_one(p1, m1, v1, hi_bf(g2[u]) * grad_scale, lr, b1, b2, eps, wd, bc1, bc2);
        const u32x2 po = {__float_as_uint(p0), __float_as_uint(p1)}, mo = {__float_as_uint(m0), __float_as_uint(m1)}, vo = {__float_as_uint(v0), __float_as_uint(v1)};
        __builtin_amdgcn_raw_buffer_store_b64(mo, rm, i * 8u, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b64(vo, rv, i * 8u, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b64(po, rp, i * 8u, 0, 2);
        __builtin_amdgcn_raw_buffer_store_b32(pack2bf(p0, p1), rq, i * 4u, 0, 2);
        asm volatile("" ::: "memory");
    }
  }
}
// VILA_SFT_ADAMW_STREAM = U (0 = the grid-stride lean kernel, 2 / 4 = element pairs per thread of the stream kernel); VILA_SFT_ADAMW_LDS = bytes of
// LDS each stream block asks for (co-residency cap; default 28672)
// Blocks of the grid-stride lean kernel: ONE PER CU by default (round 6).  The kernel exists to run beside the backward's 256x256 GEMM blocks; a
// GEMM block leaves room for exactly one 4-wave block of it per CU (2 x 224-232 + 40 VGPRs per SIMD lane).  With 1024 blocks (rounds 2-5) the
// surplus blocks took over every CU a GEMM block had just left and the next GEMM block waited for them — persistent blocks live ~1 ms — so the
// "overlap" was time-slicing: wgrad 733 us against 498 with the optimizer deferred, and the step cost the same either way (CHANGELOG round 6).
// One block per CU is always resident beside a GEMM block and never in its way: 2.2 TB/s spread over the whole backward, wgrad 615 us,
// SFT step 191.6 -> 176-180 ms.  (More bytes in flight per block — the `adamw_stream_kernel` forms below — made the GEMMs slower again:
// 203-218 ms; what the backward can spare is ~2 TB/s of the HBM pipe.)  VILA_SFT_ADAMW_GRID overrides (0 / unset = CU count).
static int adamw_lean_grid() {
    static thread_local int per_dev[16] = {0};
    static int env = -2;
    if (env == -2) { const char* e = getenv("VILA_SFT_ADAMW_GRID"); env = (e && atoi(e) >= 1) ? atoi(e) : -1; }
    if (env > 0) return env;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& n = per_dev[dev & 15];
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return n;
}
static int adamw_stream_u() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_SFT_ADAMW_STREAM"); v = (e && e[0] >= '0' && e[0] <= '9') ? atoi(e) : 0; }
    return v;
}
static int adamw_stream_lds() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VILA_SFT_ADAMW_LDS"); v = (e && e[0] >= '0' && e[0] <= '9') ? atoi(e) : 28672; }
    return v;
}
int launch_adamw_lean(float* master, float* m, float* v, const bf16_t* grad, bf16_t* param, int64_t n, float lr, float b1, float b2, float eps,
                      float wd, int step, float grad_scale, hipStream_t s) {
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    const int su = adamw_stream_u();
    if (su == 2 || su == 3 || su == 4) {
        VILA_REQUIRE(((uintptr_t)master | (uintptr_t)m | (uintptr_t)v) % 8 == 0 && ((uintptr_t)grad | (uintptr_t)param) % 4 == 0, "adamw_stream: buffer alignment");
        const int64_t chunk = (int64_t)1 << 28, n2 = n >> 1;
        int64_t done = 0;
        while (done < n2) {
            const int64_t c = (n2 - done) < chunk ? (n2 - done) : chunk;
            const int64_t per = 256 * su;
            unsigned grid = (unsigned)((c + per - 1) / per);
            if (grid > (unsigned)adamw_lean_grid()) grid = (unsigned)adamw_lean_grid();
            if (su == 3) hipLaunchKernelGGL(adamw_stream_kernel<3>, dim3(grid), dim3(256), adamw_stream_lds(), s, master + 2 * done, m + 2 * done, v + 2 * done,
                                            (const uint32_t*)(grad + 2 * done), (uint32_t*)(param + 2 * done), (unsigned)c, lr, b1, b2, eps, wd, bc1, bc2, grad_scale);
            else if (su == 2) hipLaunchKernelGGL(adamw_stream_kernel<2>, dim3(grid), dim3(256), adamw_stream_lds(), s, master + 2 * done, m + 2 * done, v + 2 * done,
                                            (const uint32_t*)(grad + 2 * done), (uint32_t*)(param + 2 * done), (unsigned)c, lr, b1, b2, eps, wd, bc1, bc2, grad_scale);
            else hipLaunchKernelGGL(adamw_stream_kernel<4>, dim3(grid), dim3(256), adamw_stream_lds(), s, master + 2 * done, m + 2 * done, v + 2 * done,
                                    (const uint32_t*)(grad + 2 * done), (uint32_t*)(param + 2 * done), (unsigned)c, lr, b1, b2, eps, wd, bc1, bc2, grad_scale);
            VILA_LAUNCH_CHECK();
            done += c;
        }
        if (n & 1) {                                                  // odd tail element: the lean kernel's single-thread path on a zero-pair launch
            hipLaunchKernelGGL(adamw_lean_kernel, dim3(1), dim3(256), 0, s, master + 2 * n2, m + 2 * n2, v + 2 * n2, (const uint32_t*)(grad + 2 * n2),
                               (uint32_t*)(param + 2 * n2), 0u, lr, b1, b2, eps, wd, bc1, bc2, grad_scale, 1);
            VILA_LAUNCH_CHECK();
        }
        return 0;
    }
    VILA_REQUIRE(((uintptr_t)master | (uintptr_t)m | (uintptr_t)v) % 8 == 0 && ((uintptr_t)grad | (uintptr_t)param) % 4 == 0, "adamw_lean: buffers must be 8-B (fp32) / 4-B (bf16) aligned");
    const int64_t chunk = (int64_t)1 << 28;                          // pairs per launch: 2 GiB of fp32 state per descriptor
    int64_t done = 0;
    const int64_t n2 = n >> 1;
    do {
        const int64_t c = (n2 - done) < chunk ? (n2 - done) : chunk;
        const int last = (done + c >= n2) ? 1 : 0;
        const int gcap = adamw_lean_grid();
        int grid = (int)((c + 255) / 256 < gcap ? (c + 255) / 256 : gcap);
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL(adamw_lean_kernel, dim3(grid), dim3(256), 0, s, master + 2 * done, m + 2 * done, v + 2 * done, (const uint32_t*)(grad + 2 * done),
                           (uint32_t*)(param + 2 * done), (unsigned)c, lr, b1, b2, eps, wd, bc1, bc2, grad_scale, (last && (n & 1)) ? 1 : 0);
        VILA_LAUNCH_CHECK();
        done += c;
    } while (done < n2);
    return 0;
}
// sum of squares of a bf16 buffer into a fp32 scalar (global grad-norm for clipping): one partial per block, added in a fixed order
__global__ __launch_bounds__(256) void sumsq_kernel(const bf16_t* __restrict__ x, int64_t n, float* __restrict__ part) {
    __shared__ float scratch[4];
    float acc = 0.f;
    EW_LOOP(n) { const float v = bf2f(x[i]); acc += v * v; }
    acc = block_sum4(acc, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = acc;
}
int launch_sumsq(const bf16_t* x, int64_t n, float* out, float* scratch /*SUMSQ_PARTS floats*/, hipStream_t s) {
    VILA_REQUIRE(scratch != nullptr, "sumsq: a scratch of %d floats is required", SUMSQ_PARTS);
    const int grid = EW_GRID(n) > SUMSQ_PARTS ? SUMSQ_PARTS : (EW_GRID(n) < 1 ? 1 : EW_GRID(n));
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, x, n, scratch);
    VILA_LAUNCH_CHECK();
    hipLaunchKernelGGL(ordered_sum_kernel, dim3(1), dim3(256), 0, s, scratch, grid, out);
    VILA_LAUNCH_CHECK();
    return 0;
}
