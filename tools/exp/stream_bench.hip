// Experiment (not product code): which per-wave load pattern streams a [N][K] bf16 weight matrix fastest when a 512-thread block owns 16 rows
// and splits K over its 8 waves (the batched-decode skinny GEMM's shape)?  hipcc --offload-arch=gfx950 -O3 -o stream_bench stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline u32x4 ldg_nt(const void* p) { return __builtin_nontemporal_load((const u32x4*)p); }
__device__ inline u32x4 ldg(const void* p) { return *(const u32x4*)p; }

// PAT 0: lane (l15 = row, lg): 16 B at lg*32 and lg*32+16 of a 128-B k-block            (MFMA operand layout, what the kernel does)
// PAT 1: lane>>2 = row, (lane&3)*16 B; second load +64 B                                (64 B contiguous per row and instruction)
// PAT 2: lane>>3 = row (8 rows), (lane&7)*16 B: one instruction = 8 rows x 128 B; second load = rows +8
// PAT 3: whole wave on ONE row: 1 KB contiguous per instruction, 16 rows one after another (GEMV-like; K split over waves in 1-KB pieces)
template <int PAT, int U, bool NT>
__global__ __launch_bounds__(512) void stream_kernel(const uint16_t* W, int N, int K, int n_items, uint32_t* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nkb = K >> 6;                 // 128-B k-blocks per row
    uint32_t acc = 0;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int r0 = item * 16;
        if constexpr (PAT == 3) {
            // K bytes per row = 2K; wave w reads 1-KB pieces w, w+8, ... of each of the 16 rows
            const int npc = (2 * K) >> 10;
            for (int r = 0; r < 16; r += 2) {
                for (int pc = wave; pc < npc; pc += 8 * U) {
                    u32x4 v[U][2];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int q = pc + 8 * u; const int qq = q < npc ? q : npc - 1;
                        const char* a = (const char*)W + (size_t)(r0 + r) * 2 * K + (size_t)qq * 1024 + lane * 16;
                        v[u][0] = NT ? ldg_nt(a) : ldg(a); v[u][1] = NT ? ldg_nt(a + 2 * K) : ldg(a + 2 * K);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc ^= v[u][0][0] ^ v[u][0][1] ^ v[u][0][2] ^ v[u][0][3] ^ v[u][1][0] ^ v[u][1][1] ^ v[u][1][2] ^ v[u][1][3];
                }
            }
        } else {
            int row, off0, off1, rowadd1 = 0;
            if (PAT == 0) { row = lane & 15; off0 = (lane >> 4) * 32; off1 = off0 + 16; }
            else if (PAT == 1) { row = lane >> 2; off0 = (lane & 3) * 16; off1 = off0 + 64; }
            else { row = lane >> 3; off0 = (lane & 7) * 16; off1 = off0; rowadd1 = 8; }
            const char* b0 = (const char*)W + (size_t)(r0 + row) * 2 * K + off0;
            const char* b1 = (const char*)W + (size_t)(r0 + row + rowadd1) * 2 * K + off1;
            for (int kb = wave; kb < nkb; kb += 8 * U) {
                u32x4 v[U][2];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = kb + 8 * u; const int qq = q < nkb ? q : nkb - 1;
                    v[u][0] = NT ? ldg_nt(b0 + (size_t)qq * 128) : ldg(b0 + (size_t)qq * 128);
                    v[u][1] = NT ? ldg_nt(b1 + (size_t)qq * 128) : ldg(b1 + (size_t)qq * 128);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc ^= v[u][0][0] ^ v[u][0][1] ^ v[u][0][2] ^ v[u][0][3] ^ v[u][1][0] ^ v[u][1][1] ^ v[u][1][2] ^ v[u][1][3];
            }
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x * 512 + tid] = acc;
}

template <int PAT, int U, bool NT>
static void run(const char* name, const uint16_t* W, int N, int K, int L, uint32_t* out, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_items = N / 16;
    const size_t per = (size_t)N * K;
    for (int i = 0; i < 3; ++i) stream_kernel<PAT, U, NT><<<grid, 512>>>(W + (size_t)(i % L) * per, N, K, n_items, out);
    hipDeviceSynchronize();
    const int reps = 56;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) stream_kernel<PAT, U, NT><<<grid, 512>>>(W + (size_t)(i % L) * per, N, K, n_items, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / reps, gbs = (double)per * 2 / (us * 1e-6) / 1e9;
    printf("%-34s N=%6d K=%6d grid=%4d  %8.2f us  %8.1f GB/s\n", name, N, K, grid, us, gbs);
}

int main() {
    const int L = 28;                                        // cycle over 28 "layers" so nothing stays in the 256-MB infinity cache
    const size_t per = (size_t)18944 * 3584;
    uint16_t* W; uint32_t* out;
    hipMalloc(&W, per * 2 * L); hipMalloc(&out, 256 * 512 * 4);
    hipMemset(W, 1, per * 2 * L);
    for (int shape = 0; shape < 2; ++shape) {
        const int N = shape == 0 ? 18944 : 3584, K = shape == 0 ? 3584 : 18944;
        const int items = N / 16, grid = items < 256 ? items : 256;
#define R(P, U, NT) run<P, U, NT>("pat" #P " U" #U " nt=" #NT, W, N, K, L, out, grid)
        R(0, 4, true); R(0, 8, true); R(0, 8, false);
        R(1, 4, true); R(1, 8, true);
        R(2, 4, true); R(2, 8, true);
        R(3, 4, true); R(3, 8, true); R(3, 8, false);
        // more blocks than CUs (finer items do not exist; this only changes the block->item striding)
        run<0, 8, true>("pat0 U8 nt grid=items", W, N, K, L, out, items);
        run<3, 8, true>("pat3 U8 nt grid=items", W, N, K, L, out, items);
    }
    return 0;
}
