// Video token assembly for BasicVideoEncoder / TSPVideoEncoder:
//   llava/model/encoders/video/tsp.py:10-11,28-52   pool(x, size, dim) = view(.., -1, size, ..).mean(dim+1) over (t, h, w) in turn,
//                                                    flatten, then per pooled frame [start tokens | features | end tokens]
//   llava/model/encoders/video/basic.py:30-41        the same without pooling
// One launch writes the finished token block of one video for one pool size:
//   out row (f, r) for pooled frame f in [0, nt/pt), r in [0, n_start + (nl/ph)(nl/pw) + n_end):
//     r <  n_start           -> start_rows[r]
//     r >= n_start + n_feat  -> end_rows[r - n_start - n_feat]
//     else                   -> mean over the pt x ph x pw window of feats[t][h*nl + w][:]   (fp32 accumulate, ONE bf16 rounding;
//                               the reference rounds to the activation dtype after each of its three means)
// HBM-bound: every input byte is read once (16 B per lane), every output byte written once.
#include "kernels.h"

__global__ void video_pool_kernel(const bf16_t* __restrict__ feats, bf16_t* __restrict__ out, const bf16_t* __restrict__ start_rows,
                                  const bf16_t* __restrict__ end_rows, int nl, int C, int pt, int ph, int pw, int n_start, int n_end,
                                  int rows_per_frame, int64_t total_chunks) {
    const int c8 = C >> 3;
    const int ho_n = nl / ph, wo_n = nl / pw, n_feat = ho_n * wo_n;
    const float inv = 1.0f / (float)(pt * ph * pw);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c8;
        const int c = (int)(i % c8) * 8;
        const int f = (int)(row / rows_per_frame), r = (int)(row % rows_per_frame);
        bf16_t* dst = out + row * C + c;
        if (r < n_start) { *(u32x4*)dst = *(const u32x4*)(start_rows + (int64_t)r * C + c); continue; }
        if (r >= n_start + n_feat) { *(u32x4*)dst = *(const u32x4*)(end_rows + (int64_t)(r - n_start - n_feat) * C + c); continue; }
        const int k = r - n_start, ho = k / wo_n, wo = k % wo_n;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int dt = 0; dt < pt; ++dt)
            for (int dh = 0; dh < ph; ++dh)
                for (int dw = 0; dw < pw; ++dw) {
                    const int64_t src = ((int64_t)(f * pt + dt) * nl + (ho * ph + dh)) * nl + (wo * pw + dw);
                    const u32x4 v = *(const u32x4*)(feats + src * C + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[2 * j] += lo_bf(v[j]); acc[2 * j + 1] += hi_bf(v[j]); }
                }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack2bf(acc[2 * j] * inv, acc[2 * j + 1] * inv);
        *(u32x4*)dst = o;
    }
}

int launch_video_pool(const bf16_t* feats, bf16_t* out, int nt, int nl, int C, int pt, int ph, int pw, const bf16_t* start_rows, int n_start,
                      const bf16_t* end_rows, int n_end, hipStream_t s) {
    VILA_REQUIRE(pt > 0 && ph > 0 && pw > 0 && nt > 0 && nl > 0, "video_pool: sizes must be positive");
    // the reference's x.view(.., -1, size, ..) raises for a ragged split
    VILA_REQUIRE(nt % pt == 0 && nl % ph == 0 && nl % pw == 0,
                 "shape '[%d, %d, %d]' is invalid for pooling by (%d, %d, %d): every pooled dimension must divide evenly", nt, nl, nl, pt, ph, pw);
    VILA_REQUIRE(C % 8 == 0, "video_pool: C=%d must be a multiple of 8", C);
    VILA_REQUIRE((n_start == 0 || start_rows != nullptr) && (n_end == 0 || end_rows != nullptr), "video_pool: token rows missing");
    const int rows_per_frame = n_start + (nl / ph) * (nl / pw) + n_end;
    const int64_t total = (int64_t)(nt / pt) * rows_per_frame * (C / 8);
    if (total == 0) return 0;
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(video_pool_kernel, dim3(grid), dim3(256), 0, s, feats, out, start_rows, end_rows, nl, C, pt, ph, pw, n_start, n_end,
                       rows_per_frame, total);
    VILA_LAUNCH_CHECK();
    return 0;
}

// Adjoint of the pooling above for the SFT step (SURVEY.md §8 row a13 over row a7: TSPVideoEncoder is an nn.Module inside the autograd graph
// the reference trains, video/tsp.py:28-52 — `mean` backward spreads each pooled row's gradient evenly over its window).
//   dpooled [(nt/pt) * (nl/ph)(nl/pw), C]   gradient of the pooled FEATURE rows only (start / end token rows belong to the embedding table)
//   dfeats  [nt, nl*nl, C]                  dfeats[t][h][w] (+)= dpooled[t/pt][h/ph][w/pw] / (pt ph pw)
// A gather: every output element has exactly one source, so no atomics; `accumulate` adds into dfeats (second and later pool sizes of the
// same video, fp32 add, one bf16 rounding per pool size).  HBM-bound: each dfeats byte written once, dpooled re-read pt*ph*pw times from L2.
__global__ void video_pool_bwd_kernel(const bf16_t* __restrict__ dpooled, bf16_t* __restrict__ dfeats, int nl, int C, int pt, int ph, int pw,
                                      int accumulate, int64_t total_chunks) {
    const int c8 = C >> 3;
    const int wo_n = nl / pw, n_feat = (nl / ph) * wo_n;
    const float inv = 1.0f / (float)(pt * ph * pw);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c8;
        const int c = (int)(i % c8) * 8;
        const int t = (int)(row / (nl * nl)), hw = (int)(row % (nl * nl)), h = hw / nl, w = hw % nl;
        const int64_t src = (int64_t)(t / pt) * n_feat + (h / ph) * wo_n + (w / pw);
        const u32x4 g = *(const u32x4*)(dpooled + src * C + c);
        bf16_t* dst = dfeats + row * C + c;
        u32x4 o;
        if (accumulate) {
            const u32x4 old = *(const u32x4*)dst;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = pack2bf(lo_bf(old[j]) + lo_bf(g[j]) * inv, hi_bf(old[j]) + hi_bf(g[j]) * inv);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = pack2bf(lo_bf(g[j]) * inv, hi_bf(g[j]) * inv);
        }
        *(u32x4*)dst = o;
    }
}

int launch_video_pool_bwd(const bf16_t* dpooled, bf16_t* dfeats, int nt, int nl, int C, int pt, int ph, int pw, int accumulate, hipStream_t s) {
    VILA_REQUIRE(pt > 0 && ph > 0 && pw > 0 && nt > 0 && nl > 0, "video_pool_bwd: sizes must be positive");
    VILA_REQUIRE(nt % pt == 0 && nl % ph == 0 && nl % pw == 0,
                 "shape '[%d, %d, %d]' is invalid for pooling by (%d, %d, %d): every pooled dimension must divide evenly", nt, nl, nl, pt, ph, pw);
    VILA_REQUIRE(C % 8 == 0, "video_pool_bwd: C=%d must be a multiple of 8", C);
    const int64_t total = (int64_t)nt * nl * nl * (C / 8);
    const int grid = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(video_pool_bwd_kernel, dim3(grid), dim3(256), 0, s, dpooled, dfeats, nl, C, pt, ph, pw, accumulate, total);
    VILA_LAUNCH_CHECK();
    return 0;
}
