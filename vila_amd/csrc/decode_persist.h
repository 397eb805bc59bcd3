// Persistent batch-1 decode token (decode_persist.hip): kernel arguments + launcher.
#pragma once
#include "common.h"

#define DP_WORKERS 7            // worker waves per block; wave 7 is the sync wave
#define DP_THREADS 512
#define DP_MAX_LAYERS 48        // per-layer pointers travel in the kernel arguments (48 x 64 B of the 4 KB segment)
#define DP_SYNC_STRIDE 64       // u32 words between two barrier words (256 B: one memory channel each)

struct DpLayer { const bf16_t *ln1, *wqkv, *bqkv, *wo, *ln2, *wg, *wu, *wd; };
struct DpArgs {
    DpLayer layer[DP_MAX_LAYERS];
    const bf16_t* norm_w; const bf16_t* lm_head; float* logits;
    bf16_t* kcache; bf16_t* vcache; int64_t kv_layer_stride;          // elements per layer: n_slots * kv_heads * max_ctx * head_dim
    const int32_t* pos_ptr; const float* rope_cs;
    bf16_t* x0; bf16_t* x1; bf16_t* q; bf16_t* act; float* part_o; float* part_ml;
    uint32_t* sync;             // word 0: error flag; words 64 + i * DP_SYNC_STRIDE, i < 17: barrier counters / generations (zeroed per token)
    int n_layers, H, F, nq, nkv, hd, vocab, max_ctx;
    float eps, scale;
    int lds_scratch, lds_outq, lds_attn, lds_wsm;                     // LDS carve (filled by the launcher)
    // measurement hook (vila_decode_persist_trace; null in product): blocks < trace_blocks stamp s_memrealtime (100 MHz) at 5 points of every
    // phase: [phase][0] worker wave 0 starts its rows, [1] it is done, [2] every worker of the block is done, [3] the sync wave's stores are
    // drained, [4] the grid barrier opened, [5 + w] worker wave w is done.  Layout [block][n_layers * 5 + 1 phases][12].
    unsigned long long* trace; int trace_blocks;
    int skew_f;                 // gate/up groups per block moved from the odd (slower) XCDs' blocks to the even ones (VILA_DECODE_PERSIST_SKEW, default 2: measured 0 / 1 / 2 / 3 -> per-XCD arrival lags even out at 2)
};
bool decode_persist_supported(int H, int F, int nq, int nkv, int hd, int n_layers, int max_ctx, int vocab);
int decode_persist_blocks();
int launch_decode_persist(DpArgs& a, hipStream_t s);
void decode_persist_set_trace(unsigned long long* buf, int n_blocks);
