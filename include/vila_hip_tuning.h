/* vila_hip_tuning.h — tuning and test switches of libvila_hip.so.  NOT part of the drop-in boundary (include/vila_hip.h):
 * these are PROCESS-GLOBAL, not thread-safe, and exist for the A/B measurements under tools/ and for the parity tests that pin one
 * kernel variant (tests/test_gpu_ops.py).  A product binding never calls them; every switch defaults to the measured-best policy.
 * Environment equivalents read once at first use: VILA_GEMM_EX, VILA_ATTN_FWD=v1, VILA_ATTN_BWD=v1, VILA_DECODE_ATTN, VILA_DECODE_CHAIN, VILA_DECODE_PERSIST. */
#ifndef VILA_HIP_TUNING_H
#define VILA_HIP_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif
/* tuning hook for the 256x256 kernel's K-loop schedule (gemm256_kernel.h SCHED): 0 = each layout's default, 1 / 2 / 3 / 5 / 6 = that schedule,
 * 9 = ablation without DMA (timing only), 10 = the round-1 schedule (SCHED 0) */
void vila_gemm_force_sched(int sched);
/* tuning hook for the launch policy fed by a workspace: whole rounds of 256x256 tiles + K-sliced tail tiles (1 = on, default; 0 = off) */
void vila_gemm_force_hybrid(int on);
/* tuning hook for the decode step's attention (caches up to 2048 positions): 2 (default) / 1 = per-head blocks over 256-key slices with the
 * merge in the o_proj GEMV's prologue (512 / 256 o_proj blocks), 0 = one block per query head over the whole context + plain o_proj */
void vila_decode_force_attn(int mode);
/* the batch-1 decode step's kernels chained over two streams (api.hip "chained decode step": kernel i streams its weights while kernel i-1
 * finishes, then waits on its done counter): 0 = off (default: measured 10 % slower than the plain step, profiles/r04_decode_chain_ab.log), 1 = on */
void vila_decode_force_chain(int on);
/* the batch-1 decode token as ONE persistent launch (decode_persist.hip: 28 layers x 5 phases + lm_head behind fence-free grid barriers, the
 * next phase's weights streaming across every barrier) + lm_head: 1 = on where the shape is supported, 0 = the per-kernel step (prologue +
 * 5 launches per layer + lm_head; the default: the two measure within 1.5 % of each other); environment: VILA_DECODE_PERSIST=1.  Logits are
 * bit-identical either way (tests/test_gpu_model.py::test_persistent_decode_step_equals_the_launch_path). */
void vila_decode_force_persist(int on);
/* measurement hook of the persistent token kernel: `buf` = device memory for [n_blocks][n_layers * 5 + 1][12] 64-bit s_memrealtime stamps (100 MHz)
 * written by blocks < n_blocks of every following launch (tools/decode_persist_trace.py); NULL = off (default) */
void vila_decode_persist_trace(void* buf, int n_blocks);
/* tuning hook: output rows per tile of the 256-wide kernel: 0 = automatic (192 when it saves tile-times), 192, 256 */
void vila_gemm_force_bm(int bm);
/* tuning / test hook: 0 = automatic tile choice, 1 = 128x128, 2 = 128x64, 3 = 256x128, 4 = 256x256 LDS-DMA, 5 = split-K if possible,
 * 6 / 7 = 128x64 LDS-DMA ring with 4 / 3 stages, 8 = 128x128 ring with 2 stages, 11 = the K-sliced 128x64 ring of gemm_ring_splitk.hip when a
 * workspace is given (automatic for M < 512 since round 5; environment: VILA_RING_SPLITK=0 turns that off), 12 / 13 / 14 = rings 7 / 8 / 6 with the
 * PIPE 2 fragment schedule (inline-asm fragment reads retired by register-tied waits: the ks = 1 reads land under the ks = 0 MFMAs; the default for
 * every ring launch since round 5, environment: VILA_RING_PIPE=0 = plain), 15 / 16 / 17 = the same three rings with the plain schedule */
void vila_gemm_force_tile(int tile);
/* leftover rows (M = 256 k + r, 1 <= r <= 16) as an extra fragment of the last 256-row tile (gemm256_kernel.h, EX): -1 = VILA_GEMM_EX from the
 * environment (default 1), 0 = off, 1 = when it saves a round of tiles (and in every K-sliced launch), 2 = whenever the rows fit (tests) */
void vila_gemm_force_ex(int mode);
/* tile order of the 256-wide kernel (gemm256_kernel.h gemm256_tile_of): -1 = automatic (columns grouped by 4 when the grid has more than 16
 * row tiles), 0 = row-tile-fastest everywhere (rounds 1-2), n > 1 = groups of n columns */
void vila_gemm_force_group(int grp);
/* the K-sliced GEMMs' reduce takes the next block's LayerNorm / RMSNorm along (prefill down_proj -> next input_layernorm, tower fc2 -> next
 * layer_norm1): 1 = on (default), 0 = separate norm launches (A/B and the parity test of the fused kernel); environment: VILA_FUSE_NORM */
void vila_gemm_force_fuse_norm(int on);
/* LayerNorm / RMSNorm over rows wider than 1536 columns: every load (x, w, b) requested up front instead of x -> reduce -> w (elementwise.hip
 * norm_block_lat_kernel; bit-identical outputs; on by default since round 5): -1 = VILA_NORM_LAT from the environment (default 1), 0 = off, 1 = on */
void vila_norm_force_lat(int on);
/* prefill (round 6): q/k/v projection K-sliced with bias + RoPE + KV-cache scatter in its reduce (instead of ring GEMM + rope_kv_kernel), and o_proj
 * K-sliced with the post-attention RMSNorm in its reduce (instead of ring GEMM + norm launch); both only where the grid is K-sliced at all
 * (>= 512 rows).  -1 = the environment's choice (VILA_PREFILL_QKV_SPLITK / VILA_PREFILL_OPROJ_SPLITK, default on), 0 = off, 1 = on */
void vila_prefill_force_fusions(int qkv_rope, int oproj_norm);
#ifdef __cplusplus
}
#endif
#endif
