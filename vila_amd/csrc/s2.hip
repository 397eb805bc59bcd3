// dynamic_s2 feature merge: everything the reference does between the vision tower and the projector
// in the dynamic_s2 branch of encode_images (llava/model/llava_arch.py:298-379) as ONE gather kernel:
//   merge_chessboard per scale  ->  F.interpolate(mode="area", fp32) to the grid of scale `resize_output_to_scale_idx` (the last scale's in
//   every shipped recipe; any scale since round 4)  ->  channel concat  ->  split_chessboard into the image's obh x obw OUTPUT blocks
//   (= bh x bw when the output grid is the last scale's, s x s for an earlier scale of s x s tiles)  ->  "b c h w -> b (h w) c"
// Descriptor words 1 / 2 carry bh | obh << 16 and bw | obw << 16 (obh = 0: the output grid is the last scale's — what rounds 1-3 wrote).
// For output block (i, j), position (y, x), scale k the value is the fp32 mean over the adaptive-average-pool window
//   rows [floor(Y*Hk/H), ceil((Y+1)*Hk/H)),  cols [floor(X*Wk/W), ceil((X+1)*Wk/W)),   Y = i*g + y, X = j*g + x
// of the scale-k chessboard, whose pixel (yy, xx) is token (yy%g)*g + xx%g of tile (yy/g)*splits + xx/g.  HBM-bound gather.
#include "kernels.h"

struct S2Args {
    const bf16_t* feats;       // [n_tiles][g*g][C]
    bf16_t* out;               // [n_blocks][g*g][n_scales*C]
    const int32_t* desc;       // device [n_blocks][6] = {tile_base, bh, bw, i, j, single}
    int n_blocks, g, C, n_scales;
    int splits[4];             // scales[k] / scales[0] for k < n_scales - 1
};

__global__ void s2_merge_kernel(S2Args p) {
    const int g = p.g, N = g * g, c8 = p.C >> 3, ns = p.n_scales;
    const int64_t total = (int64_t)p.n_blocks * N * ns * c8;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx;
        const int ch = (int)(r % c8); r /= c8;
        const int k = (int)(r % ns); r /= ns;
        const int pos = (int)(r % N); r /= N;
        const int b = (int)r;
        const int32_t* d = p.desc + b * 6;
        const int base = d[0], bh = d[1] & 0xffff, bw = d[2] & 0xffff, bi = d[3], bj = d[4], single = d[5];
        const int obh = (d[1] >> 16) ? (d[1] >> 16) : bh, obw = (d[2] >> 16) ? (d[2] >> 16) : bw;
        const int y = pos / g, x = pos % g;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        u32x4 o;
        if (single) {                                   // block_sizes[i] is None: the one tile repeated over the scales (:308-314)
            o = *(const u32x4*)(p.feats + ((int64_t)base * N + pos) * p.C + ch * 8);
        } else {
            int tile0 = base, sh, sw;
            for (int m = 0; m < k; ++m) tile0 += p.splits[m] * p.splits[m];
            if (k < ns - 1) { sh = sw = p.splits[k]; } else { sh = bh; sw = bw; }
            const int Hout = g * obh, Wout = g * obw, Hk = g * sh, Wk = g * sw;
            const int Y = bi * g + y, X = bj * g + x;
            const int ys = (Y * Hk) / Hout, ye = ((Y + 1) * Hk + Hout - 1) / Hout;
            const int xs = (X * Wk) / Wout, xe = ((X + 1) * Wk + Wout - 1) / Wout;
            for (int yy = ys; yy < ye; ++yy)
                for (int xx = xs; xx < xe; ++xx) {
                    const int tile = tile0 + (yy / g) * sw + (xx / g);
                    const int tok = (yy % g) * g + (xx % g);
                    const u32x4 v = *(const u32x4*)(p.feats + ((int64_t)tile * N + tok) * p.C + ch * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[2 * e] += lo_bf(v[e]); acc[2 * e + 1] += hi_bf(v[e]); }
                }
            const float cnt = (float)((ye - ys) * (xe - xs));
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack2bf(acc[2 * e] / cnt, acc[2 * e + 1] / cnt);
        }
        *(u32x4*)(p.out + (((int64_t)b * N + pos) * ns + k) * p.C + ch * 8) = o;
    }
}

int launch_s2_merge(const bf16_t* feats, bf16_t* out, const int32_t* desc, int n_blocks, int g, int C, int n_scales, const int* splits,
                    hipStream_t s) {
    VILA_REQUIRE(C % 8 == 0 && n_scales >= 1 && n_scales <= 4 && n_blocks > 0, "s2_merge: C %% 8, 1 <= n_scales <= 4");
    S2Args a{};
    a.feats = feats; a.out = out; a.desc = desc; a.n_blocks = n_blocks; a.g = g; a.C = C; a.n_scales = n_scales;
    for (int k = 0; k < n_scales - 1; ++k) a.splits[k] = splits[k];
    const int64_t total = (int64_t)n_blocks * g * g * n_scales * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(s2_merge_kernel, dim3(grid), dim3(256), 0, s, a);
    VILA_LAUNCH_CHECK();
    return 0;
}


// ---- backward of the merge (SURVEY.md §8f row 1, a13 for the dynamic_s2 recipe) -------------------------------------------------
// Adjoint of s2_merge_kernel as a GATHER over the tower output (no atomics, one bf16 rounding): element (tile, token, channel) of scale k
// sits at pixel (yy, xx) of that scale's chessboard; it took part, with weight 1 / |window|, in every output position (Y, X) whose
// adaptive-average-pool window [floor(Y Hk / H), ceil((Y+1) Hk / H)) x [floor(X Wk / W), ceil((X+1) Wk / W)) contains it, i.e.
//   Y in [floor(yy H / Hk), ceil((yy+1) H / Hk) - 1],  X likewise  (exact integer arithmetic; covers up- AND down-sampling scales).
// tdesc = device [n_tiles][8] {first output block of the image, bh, bw, scale k, tile row, tile col, single, 0}  (host: s2_plan)
struct S2BwdArgs {
    const bf16_t* dy;          // [n_blocks][g*g][n_scales*C]
    bf16_t* dx;                // [n_tiles][g*g][C]
    const int32_t* tdesc;
    int n_tiles, g, C, n_scales;
    int splits[4];
};

__global__ void s2_merge_bwd_kernel(S2BwdArgs p) {
    const int g = p.g, N = g * g, c8 = p.C >> 3, ns = p.n_scales;
    const int64_t total = (int64_t)p.n_tiles * N * c8;
    const int64_t ostr = (int64_t)ns * p.C;                 // elements per output position
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = idx;
        const int ch = (int)(r % c8); r /= c8;
        const int tok = (int)(r % N); r /= N;
        const int tile = (int)r;
        const int32_t* d = p.tdesc + tile * 8;
        const int blk0 = d[0], bh = d[1] & 0xffff, bw = d[2] & 0xffff, k = d[3], ti = d[4], tj = d[5], single = d[6];
        const int obh = (d[1] >> 16) ? (d[1] >> 16) : bh, obw = (d[2] >> 16) ? (d[2] >> 16) : bw;      // output block grid of the image
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (single) {                                       // the one tile was copied into every scale slot: sum over the scales
            for (int m = 0; m < ns; ++m) {
                const u32x4 v = *(const u32x4*)(p.dy + ((int64_t)blk0 * N + tok) * ostr + (int64_t)m * p.C + ch * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[2 * e] += lo_bf(v[e]); acc[2 * e + 1] += hi_bf(v[e]); }
            }
        } else {
            int sh, sw;
            if (k < ns - 1) { sh = sw = p.splits[k]; } else { sh = bh; sw = bw; }
            const int Hout = g * obh, Wout = g * obw, Hk = g * sh, Wk = g * sw;
            const int yy = ti * g + tok / g, xx = tj * g + tok % g;
            const int Y0 = (yy * Hout) / Hk, Y1 = ((yy + 1) * Hout + Hk - 1) / Hk - 1;
            const int X0 = (xx * Wout) / Wk, X1 = ((xx + 1) * Wout + Wk - 1) / Wk - 1;
            for (int Y = Y0; Y <= Y1; ++Y) {
                const int ys = (Y * Hk) / Hout, ye = ((Y + 1) * Hk + Hout - 1) / Hout;
                for (int X = X0; X <= X1; ++X) {
                    const int xs = (X * Wk) / Wout, xe = ((X + 1) * Wk + Wout - 1) / Wout;
                    const float wgt = 1.f / (float)((ye - ys) * (xe - xs));
                    const int b = blk0 + (Y / g) * obw + (X / g);
                    const int pos = (Y % g) * g + (X % g);
                    const u32x4 v = *(const u32x4*)(p.dy + ((int64_t)b * N + pos) * ostr + (int64_t)k * p.C + ch * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { acc[2 * e] += lo_bf(v[e]) * wgt; acc[2 * e + 1] += hi_bf(v[e]) * wgt; }
                }
            }
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(acc[2 * e], acc[2 * e + 1]);
        *(u32x4*)(p.dx + ((int64_t)tile * N + tok) * p.C + ch * 8) = o;
    }
}

int launch_s2_merge_bwd(const bf16_t* dy, bf16_t* dx, const int32_t* tdesc, int n_tiles, int g, int C, int n_scales, const int* splits,
                        hipStream_t s) {
    VILA_REQUIRE(C % 8 == 0 && n_scales >= 1 && n_scales <= 4 && n_tiles > 0, "s2_merge_bwd: C %% 8, 1 <= n_scales <= 4");
    S2BwdArgs a{};
    a.dy = dy; a.dx = dx; a.tdesc = tdesc; a.n_tiles = n_tiles; a.g = g; a.C = C; a.n_scales = n_scales;
    for (int k = 0; k < n_scales - 1; ++k) a.splits[k] = splits[k];
    const int64_t total = (int64_t)n_tiles * g * g * (C / 8);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(s2_merge_bwd_kernel, dim3(grid), dim3(256), 0, s, a);
    VILA_LAUNCH_CHECK();
    return 0;
}
