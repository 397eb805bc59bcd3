// Standalone GEMM micro-benchmark + spot-check for libvila_hip.so (no PyTorch: a gpurun call with it costs seconds, not minutes).
//   build:  hipcc -O2 -std=c++17 tools/gemm_bench.cpp -o tools/gemm_bench -Lvila_amd/lib -lvila_hip -Wl,-rpath,'$ORIGIN/../vila_amd/lib'
//   run:    tools/gemm_bench [fwd|bwd|lay|pmc|pol|bm|race|all]
// For every shape and DMA schedule (vila_gemm_force_sched): HIP-event timing on the null stream (random uniform [-1,1) bf16 data —
// the guide's rule 25: never quote zero-filled operands), TFLOP/s, and the max error of 384 sampled outputs against a double-precision
// dot product on the host, relative to sqrt(K) (the scale of the sum).  Layout flags: a_cm / b_cm = operand stored [K][rows].
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/vila_hip.h"
#include "../include/vila_hip_tuning.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint32_t rnd32() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 32); }
static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

struct Mat { std::vector<uint16_t> h; uint16_t* d = nullptr; int64_t rows, cols; };
static Mat make(int64_t rows, int64_t cols, float scale) {
    Mat m; m.rows = rows; m.cols = cols; m.h.resize((size_t)rows * cols);
    for (auto& x : m.h) x = f2bf(((float)(rnd32() >> 8) / 8388608.0f - 1.0f) * scale);
    CK(hipMalloc(&m.d, m.h.size() * 2));
    CK(hipMemcpy(m.d, m.h.data(), m.h.size() * 2, hipMemcpyHostToDevice));
    return m;
}

struct Case { const char* name; int M, N, K, a_cm, b_cm, residual; };
// cold mode ("precold"): the weight operand cycles over enough copies (> 600 MB) that no launch finds it in L2 or the infinity cache — the
// situation of a tower / prefill GEMM, which reads each weight once per forward.  The warm numbers of the other modes flatter short-K shapes.
static int g_cold = 0;

static void run_case(const Case& c, const std::vector<int>& scheds, void* ws, size_t ws_bytes) {
    const float sc = 1.0f;
    // stored shapes: CC [rows][K]; CM [K][rows]
    Mat A = c.a_cm ? make(c.K, c.M, sc) : make(c.M, c.K, sc);
    Mat W = c.b_cm ? make(c.K, c.N, sc) : make(c.N, c.K, sc);
    Mat R = make(c.residual ? c.M : 1, c.residual ? c.N : 8, sc);
    uint16_t* C; CK(hipMalloc(&C, (size_t)c.M * c.N * 2));
    std::vector<uint16_t> hc((size_t)c.M * c.N);
    const int64_t lda = c.a_cm ? c.M : c.K, ldw = c.b_cm ? c.N : c.K;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t wel = (size_t)c.N * c.K;
    const int copies = g_cold ? (int)((600u << 20) / (wel * 2) + 2) : 1;
    uint16_t* Wc = W.d;
    if (copies > 1) {
        CK(hipMalloc(&Wc, wel * 2 * copies));
        for (int i = 0; i < copies; ++i) CK(hipMemcpy(Wc + (size_t)i * wel, W.d, wel * 2, hipMemcpyDeviceToDevice));
    }
    int ncall = 0;
    for (int sched : scheds) {
        if (sched == 9 && (c.a_cm || c.b_cm)) continue;
        // pseudo schedules 100 / 101: default kernels with the launch policy "whole rounds + K-sliced tail tiles" off / on, automatic tile choice
        // pseudo schedules 256 / 192: default kernels with that tile height forced (0 elsewhere = automatic)
        const bool tilef = sched >= 300 && sched < 320;          // pseudo schedules 300 + t: default kernels with vila_gemm_force_tile(t)
        const bool grpf = sched >= 400 && sched < 420;            // pseudo schedules 400 + g: default kernels with vila_gemm_force_group(g) (0 = row-tile-fastest order)
        vila_gemm_force_group(grpf ? sched - 400 : -1);
        const bool bmf = sched == 256 || sched == 192;
        vila_gemm_force_bm(bmf ? sched : 0);
        const bool pol = sched >= 100 && !bmf && !tilef && !grpf;
        vila_gemm_force_sched((pol || bmf || tilef || grpf) ? 0 : sched);
        vila_gemm_force_hybrid(pol ? sched == 101 : 1);
        auto call = [&]() {
            const uint16_t* w = Wc + (size_t)(ncall++ % copies) * wel;
            int rc = vila_gemm_bf16_t(A.d, lda, c.a_cm, w, ldw, c.b_cm, nullptr, c.residual ? R.d : nullptr, c.N, C, c.N, c.M, c.N, c.K, ws, ws_bytes, nullptr);
            if (rc != 0) { fprintf(stderr, "  %s sched %d: rc=%d %s\n", c.name, sched, rc, vila_last_error()); exit(3); }
        };
        vila_gemm_force_tile(tilef ? sched - 300 : (!c.a_cm && !c.b_cm && !pol && !grpf) ? 4 : 0);          // forward layout: pin the 256x256 kernel so the schedules are comparable
        CK(hipMemset(C, 0xff, (size_t)c.M * c.N * 2));
        for (int i = 0; i < 3; ++i) call();
        CK(hipDeviceSynchronize());
        const int iters = 20;
        CK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) call();
        CK(hipEventRecord(e1, nullptr));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters, tf = 2.0 * c.M * c.N * c.K / (us * 1e-6) / 1e12;
        double worst = 0.0;
        if (sched != 9) {
            CK(hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost));
            for (int s = 0; s < 384; ++s) {
                int m = rnd32() % c.M, n = rnd32() % c.N;
                if (s < 8) { m = (s & 1) ? c.M - 1 : 0; n = (s & 2) ? c.N - 1 : 0; }             // corners: row / column tails
                if (s >= 8 && s < 16) { m = c.M - 1 - (rnd32() % 8); n = c.N - 1 - (rnd32() % 8); }
                double acc = 0.0;
                for (int k = 0; k < c.K; ++k) {
                    const float a = c.a_cm ? bf2f(A.h[(size_t)k * c.M + m]) : bf2f(A.h[(size_t)m * c.K + k]);
                    const float b = c.b_cm ? bf2f(W.h[(size_t)k * c.N + n]) : bf2f(W.h[(size_t)n * c.K + k]);
                    acc += (double)a * b;
                }
                if (c.residual) acc += bf2f(R.h[(size_t)m * c.N + n]);
                const double got = bf2f(hc[(size_t)m * c.N + n]);
                const double err = fabs(got - acc) / (sqrt((double)c.K) * sc * sc * 0.33 + fabs(acc));
                if (err > worst) worst = err;
            }
        }
        printf("%-34s M=%6d N=%6d K=%6d cm=%d%d res=%d sched=%d : %9.1f us %8.1f TF/s   err %.2e %s\n", c.name, c.M, c.N, c.K, c.a_cm, c.b_cm,
               c.residual, sched, us, tf, worst, (sched != 9 && worst > 8e-3) ? "  <-- MISMATCH" : "");
        fflush(stdout);
    }
    vila_gemm_force_tile(0);
    vila_gemm_force_sched(0);
    vila_gemm_force_hybrid(1);
    vila_gemm_force_bm(0);
    vila_gemm_force_group(-1);
    if (copies > 1) CK(hipFree(Wc));
    CK(hipFree(A.d)); CK(hipFree(W.d)); CK(hipFree(R.d)); CK(hipFree(C));
}

// Race screen for the barrier-staggered schedules: the default schedule must reproduce the lock-step round-1 schedule BIT FOR BIT (every
// accumulator receives the same MFMAs in the same order in both), on every one of `runs` launches — a hazard in the staggered groups'
// LDS traffic would show up as occasional differing tiles.  K-sliced launch policies are switched off (they change the summation order).
static int race_case(const Case& c, int runs, void* ws, size_t ws_bytes) {
    Mat A = c.a_cm ? make(c.K, c.M, 1.0f) : make(c.M, c.K, 1.0f);
    Mat W = c.b_cm ? make(c.K, c.N, 1.0f) : make(c.N, c.K, 1.0f);
    Mat R = make(c.residual ? c.M : 1, c.residual ? c.N : 8, 1.0f);
    const size_t n = (size_t)c.M * c.N;
    uint16_t* C; CK(hipMalloc(&C, n * 2));
    std::vector<uint16_t> ref(n), got(n);
    const int64_t lda = c.a_cm ? c.M : c.K, ldw = c.b_cm ? c.N : c.K;
    const bool cc = !c.a_cm && !c.b_cm;
    auto call = [&]() {
        int rc = vila_gemm_bf16_t(A.d, lda, c.a_cm, W.d, ldw, c.b_cm, nullptr, c.residual ? R.d : nullptr, c.N, C, c.N, c.M, c.N, c.K, ws, ws_bytes, nullptr);
        if (rc != 0) { fprintf(stderr, "  %s: rc=%d %s\n", c.name, rc, vila_last_error()); exit(3); }
    };
    vila_gemm_force_hybrid(0);
    vila_gemm_force_tile(cc ? 4 : 0);
    vila_gemm_force_sched(cc ? 10 : 1);                    // lock-step schedule (SCHED 0)
    CK(hipMemset(C, 0xff, n * 2)); call(); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), C, n * 2, hipMemcpyDeviceToHost));
    vila_gemm_force_sched(0);
    int bad_runs = 0; size_t bad_elems = 0;
    for (int r = 0; r < runs; ++r) {
        CK(hipMemset(C, 0xff, n * 2)); call(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), C, n * 2, hipMemcpyDeviceToHost));
        if (memcmp(got.data(), ref.data(), n * 2) != 0) {
            ++bad_runs;
            for (size_t i = 0; i < n; ++i) bad_elems += got[i] != ref[i];
        }
    }
    printf("race %-28s M=%6d N=%6d K=%6d cm=%d%d res=%d : %d launches, %d differ from the lock-step schedule (%zu elements)%s\n", c.name, c.M, c.N, c.K,
           c.a_cm, c.b_cm, c.residual, runs, bad_runs, bad_elems, bad_runs ? "  <-- MISMATCH" : "");
    fflush(stdout);
    vila_gemm_force_tile(0); vila_gemm_force_sched(0); vila_gemm_force_hybrid(1);
    CK(hipFree(A.d)); CK(hipFree(W.d)); CK(hipFree(R.d)); CK(hipFree(C));
    return bad_runs;
}

int main(int argc, char** argv) {
    const char* what = argc > 1 ? argv[1] : "all";
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("%s, %d CUs\n", prop.name, prop.multiProcessorCount);
    void* ws = nullptr; const size_t ws_bytes = (size_t)512 << 20; CK(hipMalloc(&ws, ws_bytes));
    const int T = 3076;
    std::vector<Case> fwd = {
        {"fwd qkv (SFT)", T, 4608, 3584, 0, 0, 0}, {"fwd o_proj+res (SFT)", T, 3584, 3584, 0, 0, 1}, {"fwd gate (SFT)", T, 18944, 3584, 0, 0, 0},
        {"fwd down+res (SFT)", T, 3584, 18944, 0, 0, 1}, {"square 4096", 4096, 4096, 4096, 0, 0, 0}, {"square 8192", 8192, 8192, 8192, 0, 0, 0},
        {"lm_head rows (SFT)", 1024, 152064, 3584, 0, 0, 0},
    };
    std::vector<Case> bwd = {
        {"dgrad qkv   dX=dY.W", T, 3584, 4608, 0, 1, 0}, {"dgrad o     dX=dY.W", T, 3584, 3584, 0, 1, 0}, {"dgrad gate  dX=dY.W", T, 3584, 18944, 0, 1, 0},
        {"dgrad up +res", T, 3584, 18944, 0, 1, 1}, {"dgrad down  dX=dY.W", T, 18944, 3584, 0, 1, 0},
        {"wgrad qkv   dW=dY^T.X", 4608, 3584, T, 1, 1, 0}, {"wgrad o     dW=dY^T.X", 3584, 3584, T, 1, 1, 0}, {"wgrad gate  dW=dY^T.X", 18944, 3584, T, 1, 1, 0},
        {"wgrad down  dW=dY^T.X", 3584, 18944, T, 1, 1, 0}, {"wgrad lm_head", 152064, 3584, 1024, 1, 1, 0}, {"dgrad lm_head (split-K)", 1024, 3584, 152064, 0, 1, 0},
        {"a_cm only (coverage)", 4096, 4096, 4096, 1, 0, 0}, {"ViT dgrad fc2 (split-K)", 4096, 4304, 1152, 0, 1, 0}, {"ViT wgrad fc1 (split-K)", 4304, 1152, 4096, 1, 1, 0},
        {"ragged: M tail, K tail", 1000, 1032, 1496, 0, 1, 0}, {"ragged wgrad", 1032, 520, 777, 1, 1, 1},
    };
    std::vector<Case> lay = {
        {"layouts 4096^3 cc", 4096, 4096, 4096, 0, 0, 0}, {"layouts 4096^3 a_cm", 4096, 4096, 4096, 1, 0, 0}, {"layouts 4096^3 b_cm", 4096, 4096, 4096, 0, 1, 0},
        {"layouts 4096^3 both", 4096, 4096, 4096, 1, 1, 0},
        {"wgrad gate as a_cm + X^T", 18944, 3584, 3136, 1, 0, 0}, {"wgrad gate both cm", 18944, 3584, 3076, 1, 1, 0}, {"wgrad gate cc (old path)", 18944, 3584, 3136, 0, 0, 0},
        {"dgrad gate b_cm", 3076, 3584, 18944, 0, 1, 0}, {"dgrad gate cc (old path)", 3076, 3584, 18944, 0, 0, 0},
        {"dgrad down b_cm", 3076, 18944, 3584, 0, 1, 0}, {"dgrad down cc (old path)", 3076, 18944, 3584, 0, 0, 0},
    };
    if (!strcmp(what, "pmc")) {        // counter runs (rocprofv3 --pmc): ONE shape per kernel name, so per-kernel sums are comparable
        run_case(lay[0], {0, 5, 10}, ws, ws_bytes);
        run_case(lay[2], {0, 5}, ws, ws_bytes);
        run_case(lay[3], {0, 5}, ws, ws_bytes);
    }
    if (!strcmp(what, "pol")) {        // launch policies off / on, the SFT shapes they target
        for (int i : {3, 1}) run_case(fwd[i], {100, 101, 100, 101}, ws, ws_bytes);
        for (int i : {2, 3, 4, 7, 8, 6, 5}) run_case(bwd[i], {100, 101, 100, 101}, ws, ws_bytes);
    }
    if (!strcmp(what, "bm")) {         // 256- vs 192-row tiles on the shapes with M = 3076
        for (int i : {0, 1, 2, 3, 4}) run_case(fwd[i], {256, 192, 256, 192}, ws, ws_bytes);
        for (int i : {0, 1, 2, 3, 4, 14}) run_case(bwd[i], {256, 192, 256, 192}, ws, ws_bytes);
    }
    if (!strcmp(what, "pre")) {        // prefill / tower shapes: automatic choice vs forced kernels (5 = split-K 256^2, 4 = 256^2, 7 = 128x64 ring, 8 = 128x128 ring)
        std::vector<Case> pc = {
            {"LLM qkv  S=768", 768, 4608, 3584, 0, 0, 0}, {"LLM o+res S=768", 768, 3584, 3584, 0, 0, 1},
            {"ViT qkv  M=1024", 1024, 3456, 1152, 0, 0, 0}, {"ViT out+res", 1024, 1152, 1152, 0, 0, 1}, {"ViT fc2+res", 1024, 1152, 4304, 0, 0, 1},
        };
        for (auto& c : pc) run_case(c, {300, 305, 304, 307, 308, 300, 305, 304, 307, 308}, ws, ws_bytes);
    }
    if (!strcmp(what, "presmall")) {   // short prompts (text-only chat, one image + a short question), cold weights: automatic choice vs split-K 256^2 vs the 128x64 ring
        g_cold = 1;
        std::vector<Case> pc = {
            {"LLM down+res M=64", 64, 3584, 18944, 0, 0, 1}, {"LLM down+res M=160", 160, 3584, 18944, 0, 0, 1}, {"LLM down+res M=289", 289, 3584, 18944, 0, 0, 1},
            {"LLM qkv M=64", 64, 4608, 3584, 0, 0, 0}, {"LLM qkv M=289", 289, 4608, 3584, 0, 0, 0}, {"LLM o+res M=64", 64, 3584, 3584, 0, 0, 1}, {"LLM o+res M=289", 289, 3584, 3584, 0, 0, 1},
        };
        for (auto& c : pc) run_case(c, {300, 305, 307, 300, 305, 307}, ws, ws_bytes);
        g_cold = 0;
    }
    if (!strcmp(what, "precold")) {    // the same question with COLD weights (what a forward pass sees), one and eight images, S = 769 prefill
        g_cold = 1;
        std::vector<Case> pc = {
            {"ViT qkv  M=1024 cold", 1024, 3456, 1152, 0, 0, 0}, {"ViT out+res cold", 1024, 1152, 1152, 0, 0, 1}, {"ViT fc1 cold", 1024, 4304, 1152, 0, 0, 0},
            {"ViT fc2+res cold", 1024, 1152, 4304, 0, 0, 1},
            {"LLM qkv  S=769 cold", 769, 4608, 3584, 0, 0, 0}, {"LLM o+res S=769 cold", 769, 3584, 3584, 0, 0, 1},
            {"ViT qkv  M=8192 cold", 8192, 3456, 1152, 0, 0, 0}, {"ViT fc1 M=8192 cold", 8192, 4304, 1152, 0, 0, 0}, {"ViT fc2 M=8192 cold", 8192, 1152, 4304, 0, 0, 1},
        };
        for (auto& c : pc) run_case(c, {300, 305, 304, 307, 308, 300, 305, 304, 307, 308}, ws, ws_bytes);
        g_cold = 0;
    }
    if (!strcmp(what, "prering")) {    // ring kernels, cold weights: the automatic choice (300) vs the 128x64 3-stage ring / the 128x128 2-stage ring with the PIPE 2
        g_cold = 1;                    // fragment schedule (312 / 313) and with the plain one (315 / 316).  (The round-5 run of this mode, with the since-removed
        std::vector<Case> pc = {       // PIPE 1 and 128x128 3- / 4-stage variants in it, is profiles/r05_gemm_bench_prering.log.)
            {"LLM qkv  S=769 cold", 769, 4608, 3584, 0, 0, 0}, {"LLM o+res S=769 cold", 769, 3584, 3584, 0, 0, 1},
            {"ViT qkv  M=1024 cold", 1024, 3456, 1152, 0, 0, 0}, {"ViT out+res cold", 1024, 1152, 1152, 0, 0, 1}, {"ViT fc1 cold", 1024, 4304, 1152, 0, 0, 0},
            {"LLM qkv M=289 cold", 289, 4608, 3584, 0, 0, 0}, {"LLM o+res M=289 cold", 289, 3584, 3584, 0, 0, 1},
        };
        for (auto& c : pc) run_case(c, {300, 312, 315, 313, 316, 300, 312, 315, 313, 316}, ws, ws_bytes);
        // short prompts: the K-sliced 128x64 ring (force_tile 11, gemm_ring_splitk.hip) vs the automatic choice and the plain ring
        std::vector<Case> sc = {
            {"LLM qkv M=64 cold", 64, 4608, 3584, 0, 0, 0}, {"LLM qkv M=160 cold", 160, 4608, 3584, 0, 0, 0}, {"LLM qkv M=289 cold", 289, 4608, 3584, 0, 0, 0},
            {"LLM o+res M=64 cold", 64, 3584, 3584, 0, 0, 1}, {"LLM o+res M=289 cold", 289, 3584, 3584, 0, 0, 1},
            {"Lite qkv M=154 cold", 154, 2560, 2048, 0, 0, 0}, {"Lite o+res M=154 cold", 154, 2048, 2048, 0, 0, 1},
        };
        for (auto& c : sc) run_case(c, {300, 307, 311, 300, 307, 311}, ws, ws_bytes);
        g_cold = 0;
    }
    if (!strcmp(what, "grp")) {        // tile order: row-tile-fastest strips (400) vs columns grouped by 4 / 2 / 8 (8 x 4 patches per XCD)
        for (int i : {5, 6, 7, 8, 9, 0, 2, 4}) run_case(bwd[i], {400, 404, 402, 408, 400, 404}, ws, ws_bytes);
        for (int i : {2, 3, 5, 6}) run_case(fwd[i], {400, 404, 400, 404}, ws, ws_bytes);
    }
    if (!strcmp(what, "one")) {        // ONE backward shape, default kernels only: the unit of a --pmc pass (tools/pmc_gemm_sft.sh); 23 GEMM calls
        const int idx = argc > 2 ? atoi(argv[2]) : 7;
        if (idx < 0 || idx >= (int)bwd.size()) { fprintf(stderr, "one: index 0..%d\n", (int)bwd.size() - 1); return 2; }
        run_case(bwd[idx], {0}, ws, ws_bytes);
    }
    if (!strcmp(what, "race")) {
        std::vector<Case> rc = {
            {"256^3", 256, 256, 256, 0, 0, 0}, {"512^3", 512, 512, 512, 0, 0, 0}, {"4096^3", 4096, 4096, 4096, 0, 0, 0},
            {"fwd down+res", 3076, 3584, 18944, 0, 0, 1}, {"fwd o_proj (192-row tiles)", 3076, 3584, 3584, 0, 0, 1}, {"ragged cc", 1000, 1032, 1496, 0, 0, 1},
            {"dgrad gate", 3076, 3584, 18944, 0, 1, 0}, {"dgrad ragged", 1000, 1032, 1496, 0, 1, 1},
            {"wgrad qkv", 4608, 3584, 3076, 1, 1, 0}, {"wgrad ragged", 1032, 520, 777, 1, 1, 1}, {"a_cm only", 2048, 2048, 2048, 1, 0, 0},
        };
        int bad = 0;
        for (auto& c : rc) bad += race_case(c, c.M * (int64_t)c.N > 8000000 ? 30 : 100, ws, ws_bytes);
        printf("race screen: %s\n", bad ? "FAILED" : "clean");
    }
    if (!strcmp(what, "lay")) for (auto& c : lay) run_case(c, {0, 1}, ws, ws_bytes);
    if (!strcmp(what, "fwd") || !strcmp(what, "all")) for (auto& c : fwd) run_case(c, {0, 6, 10, 0, 6, 10}, ws, ws_bytes);     // default (SCHED 7) vs the 8-barrier role split vs round 1
    if (!strcmp(what, "bwd") || !strcmp(what, "all")) for (auto& c : bwd) run_case(c, {0, 2, 0, 2}, ws, ws_bytes);            // default (SCHED 7 + launch policies) vs the lock-step two-tiles-ahead schedule     // contraction-major: 0 = default (two tiles ahead), 1 = one tile ahead
    return 0;
}
