// Instantiations of the 256x256 kernel for contraction-major operands (dgrad / wgrad without transposed copies) and for the DMA
// schedule variants (gemm256_kernel.h SCHED), kept in their own translation unit so the two files compile in parallel.
//
// Measured on MI355X (tools/gemm_bench, uniform [-1,1) operands; logs under profiles/r02_gemm_bench_*.log), TF/s, T = 3076:
//   schedule                          fwd qkv / o / gate / down      dgrad qkv / o / gate / down     wgrad qkv / o / gate / down
//   0  lock-step, one tile ahead        917 /  700 /  896 /  796      646 /  630 /  653 /  790       804 /  673 /  722 /  747
//   2  lock-step, two tiles ahead       no change for the forward      737 /  719 /  773 /  865       905 /  760 /  810 /  832
//   6  role split, 8 barriers / tile    1014 /  780 / 1026 /  883      761 /  738 /  807 /  906       -8 % vs SCHED 2
//   7  role split, 4 barriers / tile    1105 /  882 / 1117 /  969      845 /  843 /  888 /  955       983 /  813 /  830 /  839   <- default
//   (wgrad gate / down with the whole-rounds + sliced-tail policy of gemm256.hip: 870 / 889)
#include "gemm256_kernel.h"

extern int g_gemm256_sched;

template <bool ACM, bool BCM>
static int launch_cm_t(const GemmArgs& a, int sched, hipStream_t s) {
    // default for every layout: the role-split schedule with two 32-MFMA phases per K-tile (SCHED 7).  vila_gemm_force_sched:
    // 1 = SCHED 0 (round 1), 2 / 5 / 6 = that schedule
    if (sched == 1) return launch256_t<0, EPI_NONE, ACM, BCM, 0>(a, s);
    if (sched == 2) return launch256_t<0, EPI_NONE, ACM, BCM, 2>(a, s);
    if (sched == 5) return launch256_t<0, EPI_NONE, ACM, BCM, 5>(a, s);
    if (sched == 6) return launch256_t<0, EPI_NONE, ACM, BCM, 6>(a, s);
    // (192-row tiles were measured for dgrad too: -2 % on K = 3584 / 4608, +3 % on K = 18944: not used for contraction-major operands)
    return launch256_t<0, EPI_NONE, ACM, BCM, 7>(a, s);
}

// bf16 out (+ bias / residual), no activation: dX = dY . W (b_cm) and dW = dY^T . X (a_cm, b_cm)
int launch_gemm256_cm(const GemmArgs& a, hipStream_t s) {
    VILA_REQUIRE(a.epi == EPI_NONE && !a.out_f32, "gemm256: contraction-major operands support only the plain bf16 epilogue");
    if (a.a_cm && a.b_cm) return launch_cm_t<true, true>(a, g_gemm256_sched, s);
    if (a.b_cm) return launch_cm_t<false, true>(a, g_gemm256_sched, s);
    return launch_cm_t<true, false>(a, g_gemm256_sched, s);
}

int launch_gemm256_cm_splitk(const GemmArgs& b, int splits, float* slab, int per, hipStream_t s) {
    (void)slab;
    if (b.a_cm && b.b_cm) return launch256_t<3, EPI_NONE, true, true, 7>(b, s, splits, 0, -1, 0, per);
    if (b.b_cm) return launch256_t<3, EPI_NONE, false, true, 7>(b, s, splits, 0, -1, 0, per);
    return launch256_t<3, EPI_NONE, true, false, 7>(b, s, splits, 0, -1, 0, per);
}

// a tile range of a contraction-major GEMM: mode 0 = finished bf16 tiles, mode 5 = K-sliced raw sums into compact per-tile slabs
// (gemm256.hip try_hybrid: whole rounds + sliced tail)
int launch_gemm256_cm_range(const GemmArgs& a, int mode, int splits, int tile0, int n_tiles, int per, hipStream_t s) {
    if (mode == 0) {
        if (a.a_cm && a.b_cm) return launch256_t<0, EPI_NONE, true, true, 7>(a, s, 1, tile0, n_tiles);
        if (a.b_cm) return launch256_t<0, EPI_NONE, false, true, 7>(a, s, 1, tile0, n_tiles);
        return launch256_t<0, EPI_NONE, true, false, 7>(a, s, 1, tile0, n_tiles);
    }
    if (a.a_cm && a.b_cm) return launch256_t<5, EPI_NONE, true, true, 7>(a, s, splits, tile0, n_tiles, 0, per);
    if (a.b_cm) return launch256_t<5, EPI_NONE, false, true, 7>(a, s, splits, tile0, n_tiles, 0, per);
    return launch256_t<5, EPI_NONE, true, false, 7>(a, s, splits, tile0, n_tiles, 0, per);
}

// forward layout with another DMA schedule (tuning / A-B measurement through vila_gemm_force_sched)
int launch_gemm256_sched(const GemmArgs& a, int sched, hipStream_t s) {
    switch (sched) {
        case 1: return launch256_t<0, EPI_NONE, false, false, 1>(a, s);
        case 2: return launch256_t<0, EPI_NONE, false, false, 2>(a, s);
        case 5: return launch256_t<0, EPI_NONE, false, false, 5>(a, s);
        case 6: return launch256_t<0, EPI_NONE, false, false, 6>(a, s);
        case 9: return launch256_t<0, EPI_NONE, false, false, 9>(a, s);
        case 3: return launch256_t<0, EPI_NONE, false, false, 3>(a, s);
        case 10: return launch256_t<0, EPI_NONE, false, false, 0>(a, s);     // round-1 schedule (one tile ahead, fragments read per phase)
        default: return launch256_t<0, EPI_NONE, false, false, T256_CC_SCHED>(a, s);
    }
}
