// Stochastic next-token choice for generate(do_sample=True) — the reference's server default (server.py:101-102,185-187: temperature 0.2,
// top_p 0.9, do_sample = temperature > 0; HF GenerationConfig's default top_k = 50 stays in force).  HF order of operations
// (GenerationMixin._get_logits_processor + sample): logits / temperature -> TopK (keep the k largest) -> TopP (ascending cumulative
// softmax, drop while cum <= 1 - top_p, keep at least one) -> softmax over what is left -> multinomial.
//
// Everything stays on the device so that a sampled decode step replays from a hipGraph like the greedy one:
//   stage 1  256 blocks: each sorts its slice of the vocabulary (bitonic, LDS) and emits its 64 best (value, index) pairs
//   stage 2  16 blocks : 1024 candidates each -> 64 best;   stage 3  1 block: 1024 -> the global top 64 (descending), then ONE wave applies
//            temperature / top-k / top-p and draws from the renormalised set with a counter-based uniform: splitmix64(seed, *counter) —
//            `counter` is the device-resident position of the token, so every replay of the captured step draws a fresh number.
// That path serves top_k in 1..64 (64 candidates survive the selection).  Any other top_k — 0 = no top-k filter, as HF treats it, or k > 64 up to
// the vocabulary — goes through smp_large_kernel below: ONE 1024-thread block, radix selection over the 64-bit (value, index) keys for the k-th
// largest, a mass-weighted radix descent for the nucleus threshold, a draw by prefix sums in index order.  The draw cannot match
// torch.multinomial's generator bit for bit — parity for this op is distributional (tests/test_gpu_sampling.py).
#include "kernels.h"

#define SMP_CAND 64
#define SMP_S1_BLOCKS 256
#define SMP_S2_BLOCKS 16

// order-preserving key: larger float -> larger key; ties broken towards the LOWER index (index stored inverted in the low bits)
__device__ __forceinline__ uint64_t smp_key(float v, int idx) {
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | (uint32_t)(0x7fffffff - idx);
}
__device__ __forceinline__ float smp_val(uint64_t k) {
    uint32_t u = (uint32_t)(k >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}
__device__ __forceinline__ int smp_idx(uint64_t k) { return 0x7fffffff - (int)(uint32_t)(k & 0xffffffffu); }

// descending bitonic sort of 1024 keys in LDS by 256 threads
__device__ void smp_sort1024(uint64_t* keys) {
    for (int size = 2; size <= 1024; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < 512; t += 256) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
        }
    __syncthreads();
}

// stage 1 / 2: candidates in, SMP_CAND best per block out.  in_vals == nullptr: read raw logits [n] (index = position);
// otherwise read (key) candidates produced by the previous stage.
__global__ __launch_bounds__(256) void smp_select_kernel(const float* __restrict__ logits, const uint64_t* __restrict__ in_keys, int n, int per_block,
                                                         uint64_t* __restrict__ out_keys) {
    __shared__ uint64_t keys[1024];
    const int base = blockIdx.x * per_block;
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int g = base + i;
        uint64_t k = 0;                                             // below every real key
        if (i < per_block && g < n) k = (in_keys != nullptr) ? in_keys[g] : smp_key(logits[g], g);
        keys[i] = k;
    }
    smp_sort1024(keys);
    if (threadIdx.x < SMP_CAND) out_keys[blockIdx.x * SMP_CAND + threadIdx.x] = keys[threadIdx.x];
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// stage 3: global top-64, then temperature / top-k / top-p / draw by wave 0
__global__ __launch_bounds__(256) void smp_final_kernel(const uint64_t* __restrict__ in_keys, int n_in, float inv_temperature, int top_k, float top_p,
                                                        uint64_t seed_imm, const uint64_t* __restrict__ seed_dev, const int32_t* __restrict__ counter, int64_t* __restrict__ out, float* __restrict__ prob_out) {
    __shared__ uint64_t keys[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) keys[i] = i < n_in ? in_keys[i] : 0;
    smp_sort1024(keys);
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    const uint64_t k = keys[lane];
    const bool real = k != 0 && lane < top_k;                      // TopK: the k most likely survive
    const float z = smp_val(k) * inv_temperature, zmax = smp_val(keys[0]) * inv_temperature;
    const float e = real ? __expf(z - zmax) : 0.f;
    const float total = wave_sum(e);
    const float p = e / total;                                      // softmax over the top-k set (descending order by construction)
    // exclusive prefix sum over the wave (descending probabilities)
    float incl = p;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const float before = incl - p;                                  // probability mass of strictly more likely tokens
    const bool keep = real && (lane == 0 || before < top_p);        // TopP (HF keeps the token that crosses the threshold; at least one)
    const float pk = keep ? p : 0.f;
    const float kept_total = wave_sum(pk);
    float cum = pk;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(cum, o, 64);
        if (lane >= o) cum += t;
    }
    const uint64_t seed = seed_dev != nullptr ? *seed_dev : seed_imm;
    const uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)(uint32_t)(counter != nullptr ? *counter : 0)));
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f) * kept_total;      // uniform in [0, kept_total)
    const unsigned long long hit = __ballot(keep && cum > u);
    const int pick = hit ? __ffsll((long long)hit) - 1 : 0;
    if (lane == pick) *out = (int64_t)smp_idx(k);
    if (prob_out != nullptr) prob_out[lane] = pk / kept_total;     // optional: the distribution that was sampled (tests)
    if (prob_out != nullptr) ((int*)(prob_out + 64))[lane] = real ? smp_idx(k) : -1;
}

// ---- any top_k (round 4): exact selection without a candidate cut ----------------------------------------------------------------------------
// Keys are a strict total order (value, then LOWER index first), so "the k most likely" and "the tokens whose more-likely mass is below top_p"
// are both `key >= threshold` sets:
//   T_k : 8 radix passes (one byte of the 64-bit key per pass, integer histograms) find the k-th largest key
//   T_p : 8 passes with MASS histograms (sum of exp((v - vmax) / T) per bin, accumulated as 2^-40 fixed-point integers so the sums do not depend on
//         lane arrival order; one histogram row per wave, rows added in a fixed order) descend to
//         the smallest key whose strictly-more-likely mass is still below top_p x total  (HF TopPLogitsWarper: drop while ascending cumulative
//         mass <= 1 - top_p; the top token always survives)
//   draw: u in [0, kept mass) located by prefix sums over the kept tokens in INDEX order (any fixed order gives the same distribution)
#define SMPL_T 1024
__device__ __forceinline__ float smpl_block_sum(float v, float* red) {      // fixed-order tree: the same bits on every run
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < SMPL_T / 64; ++i) t += red[i];
    return t;
}
__global__ __launch_bounds__(SMPL_T) void smp_large_kernel(const float* __restrict__ logits, int n, float inv_temperature, int top_k, float top_p,
                                                           uint64_t seed_imm, const uint64_t* __restrict__ seed_dev, const int32_t* __restrict__ counter,
                                                           int64_t* __restrict__ out, float* __restrict__ prob_out) {
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long histq[SMPL_T / 64][256];     // nucleus masses in 2^-40 fixed point: INTEGER atomics commute, so a bin's sum
                                                               // has the same bits whatever order the lanes of a wave arrive in
    __shared__ float red[SMPL_T / 64];
    __shared__ float scan[SMPL_T];
    __shared__ uint64_t sh_key;
    __shared__ float sh_f;
    __shared__ int sh_i;
    const int tid = threadIdx.x, wave = tid >> 6;
    // ---- the largest key (value -> softmax shift) ----
    uint64_t kmax = 0;
    for (int i = tid; i < n; i += SMPL_T) { const uint64_t k = smp_key(logits[i], i); kmax = k > kmax ? k : kmax; }
    for (int o = 32; o > 0; o >>= 1) { const uint64_t t = __shfl_xor((unsigned long long)kmax, o, 64); kmax = t > kmax ? t : kmax; }
    if ((tid & 63) == 0) ((uint64_t*)histq)[wave] = kmax;
    __syncthreads();
    if (tid == 0) { uint64_t m = 0; for (int w = 0; w < SMPL_T / 64; ++w) { const uint64_t t = ((uint64_t*)histq)[w]; m = t > m ? t : m; } sh_key = m; }
    __syncthreads();
    kmax = sh_key;
    const float zmax = smp_val(kmax) * inv_temperature;
    __syncthreads();
    // ---- T_k: the k-th largest key (0 = keep everything) ----
    uint64_t Tk = 0;
    if (top_k >= 1 && top_k < n) {
        uint64_t prefix = 0;
        uint32_t remaining = (uint32_t)top_k;
        for (int b = 7; b >= 0; --b) {
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            const uint64_t hi_mask = b == 7 ? 0ull : (~0ull << (8 * (b + 1)));
            for (int i = tid; i < n; i += SMPL_T) {
                const uint64_t k = smp_key(logits[i], i);
                if ((k & hi_mask) == prefix) atomicAdd(&hist[(k >> (8 * b)) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                uint32_t cum = 0; int d = 255;
                for (; d > 0; --d) { if (cum + hist[d] >= remaining) break; cum += hist[d]; }
                sh_i = d; sh_f = __uint_as_float(cum);
            }
            __syncthreads();
            prefix |= (uint64_t)sh_i << (8 * b);
            remaining -= __float_as_uint(sh_f);
            __syncthreads();
        }
        Tk = prefix;
    }
    // ---- mass of the top-k set ----
    float part = 0.f;
    for (int i = tid; i < n; i += SMPL_T) { const float v = logits[i]; if (smp_key(v, i) >= Tk) part += __expf(v * inv_temperature - zmax); }
    const float total_k = smpl_block_sum(part, red);
    // ---- T_p: the smallest key kept by the nucleus ----
    uint64_t T = Tk;
    if (top_p < 1.f) {
        const float thr = top_p * total_k;
        uint64_t prefix = 0;
        float above = 0.f;                                       // mass of the keys above the current prefix range
        for (int b = 7; b >= 0; --b) {
            for (int i = tid; i < (SMPL_T / 64) * 256; i += SMPL_T) (&histq[0][0])[i] = 0ull;
            __syncthreads();
            const uint64_t hi_mask = b == 7 ? 0ull : (~0ull << (8 * (b + 1)));
            for (int i = tid; i < n; i += SMPL_T) {
                const float v = logits[i];
                const uint64_t k = smp_key(v, i);
                if (k >= Tk && (k & hi_mask) == prefix)      // e <= 1 (shifted by the largest logit): e * 2^40 fits, 2^18 of them fit 64 bits
                    atomicAdd(&histq[wave][(k >> (8 * b)) & 255u], (unsigned long long)(__expf(v * inv_temperature - zmax) * 1099511627776.0f));
            }
            __syncthreads();
            if (tid < 256) { unsigned long long m = 0ull; for (int w = 0; w < SMPL_T / 64; ++w) m += histq[w][tid]; scan[tid] = (float)m * (1.0f / 1099511627776.0f); }
            __syncthreads();
            if (tid == 0) {
                // bins from the top: G(d) = above + mass of bins > d; descend into the LOWEST bin whose G is still below the threshold
                float g = above; int d = 255, best = -1; float gbest = above;
                for (; d >= 0; --d) {
                    if (g < thr) { if (scan[d] > 0.f || best < 0) { best = d; gbest = g; } } else break;
                    g += scan[d];
                }
                if (best < 0) { best = 255; gbest = above; }
                // (only non-empty bins can hold the threshold key; the top bin of the first pass holds the largest key, which is always kept)
                sh_i = best; sh_f = gbest;
            }
            __syncthreads();
            prefix |= (uint64_t)sh_i << (8 * b);
            above = sh_f;
            __syncthreads();
        }
        T = prefix > Tk ? prefix : Tk;
    }
    // ---- draw in index order ----
    const int chunk = (n + SMPL_T - 1) / SMPL_T, i0 = tid * chunk, i1 = (i0 + chunk < n) ? i0 + chunk : n;
    float mine = 0.f;
    for (int i = i0; i < i1; ++i) { const float v = logits[i]; if (smp_key(v, i) >= T) mine += __expf(v * inv_temperature - zmax); }
    scan[tid] = mine;
    __syncthreads();
    if (tid == 0) {                                              // 1024 partials: a serial scan by one thread keeps the order fixed
        float acc = 0.f;
        for (int t = 0; t < SMPL_T; ++t) { const float m = scan[t]; scan[t] = acc; acc += m; }
        sh_f = acc;
        sh_i = -1;
    }
    __syncthreads();
    const float kept_total = sh_f;
    const uint64_t seed = seed_dev != nullptr ? *seed_dev : seed_imm;
    const uint64_t r = splitmix64(seed ^ splitmix64((uint64_t)(uint32_t)(counter != nullptr ? *counter : 0)));
    const float u = (float)(r >> 40) * (1.0f / 16777216.0f) * kept_total;
    const float base = scan[tid];
    if (mine > 0.f && u >= base && (u < base + mine || tid == SMPL_T - 1)) {
        float acc = base; int pick = -1, last = -1;
        for (int i = i0; i < i1; ++i) {
            const float v = logits[i];
            if (smp_key(v, i) >= T) { last = i; acc += __expf(v * inv_temperature - zmax); if (pick < 0 && u < acc) pick = i; }
        }
        atomicMax(&sh_i, pick >= 0 ? pick : last);               // (at most one chunk contains u; rounding at a chunk edge: the later one wins)
    }
    __syncthreads();
    if (tid == 0) *out = (int64_t)(sh_i >= 0 ? sh_i : smp_idx(kmax));      // u beyond the last partial by rounding: the most likely token
    if (prob_out != nullptr)
        for (int i = tid; i < n; i += SMPL_T) { const float v = logits[i]; prob_out[i] = smp_key(v, i) >= T ? __expf(v * inv_temperature - zmax) / kept_total : 0.f; }
}

size_t sample_workspace_bytes() { return (size_t)(SMP_S1_BLOCKS + SMP_S2_BLOCKS) * SMP_CAND * sizeof(uint64_t) + 256; }

int launch_sample(const float* logits, int n, float temperature, int top_k, float top_p, uint64_t seed, const uint64_t* seed_dev, const int32_t* counter, int64_t* out,
                  void* workspace, float* prob_out, hipStream_t s) {
    VILA_REQUIRE(n > 0 && temperature > 0.f, "sample: temperature must be positive (got %g); use greedy search for temperature 0", (double)temperature);
    VILA_REQUIRE(top_k >= 0, "sample: top_k must be >= 0 (got %d; 0 = no top-k filter)", top_k);
    VILA_REQUIRE(top_p > 0.f && top_p <= 1.f, "sample: top_p must be in (0, 1] (got %g)", (double)top_p);
    if (top_k == 0 || top_k > SMP_CAND) {                       // any k: exact radix selection in one block (prob_out: dense [n])
        hipLaunchKernelGGL(smp_large_kernel, dim3(1), dim3(SMPL_T), 0, s, logits, n, 1.0f / temperature, top_k, top_p, seed, seed_dev, counter, out, prob_out);
        VILA_LAUNCH_CHECK();
        return 0;
    }
    const int per1 = cdiv(n, SMP_S1_BLOCKS);
    VILA_REQUIRE(per1 <= 1024, "sample: vocabulary of %d exceeds %d x 1024 entries", n, SMP_S1_BLOCKS);
    uint64_t* c1 = (uint64_t*)workspace;
    uint64_t* c2 = c1 + SMP_S1_BLOCKS * SMP_CAND;
    hipLaunchKernelGGL(smp_select_kernel, dim3(SMP_S1_BLOCKS), dim3(256), 0, s, logits, (const uint64_t*)nullptr, n, per1, c1);
    VILA_LAUNCH_CHECK();
    hipLaunchKernelGGL(smp_select_kernel, dim3(SMP_S2_BLOCKS), dim3(256), 0, s, (const float*)nullptr, (const uint64_t*)c1, SMP_S1_BLOCKS * SMP_CAND, 1024, c2);
    VILA_LAUNCH_CHECK();
    hipLaunchKernelGGL(smp_final_kernel, dim3(1), dim3(256), 0, s, (const uint64_t*)c2, SMP_S2_BLOCKS * SMP_CAND, 1.0f / temperature, top_k, top_p, seed, seed_dev, counter, out, prob_out);
    VILA_LAUNCH_CHECK();
    return 0;
}
