// Instantiations of the 256x256 kernel for contraction-major operands (dgrad / wgrad without transposed copies) and for the DMA
// schedule variants (gemm256_kernel.h SCHED), kept in their own translation unit so the two files compile in parallel.
#include "gemm256_kernel.h"

extern int g_gemm256_sched;

template <bool ACM, bool BCM>
static int launch_cm_t(const GemmArgs& a, int sched, hipStream_t s) {
    switch (sched) {
        case 1: return launch256_t<0, EPI_NONE, ACM, BCM, 1>(a, s);
        case 2: return launch256_t<0, EPI_NONE, ACM, BCM, 2>(a, s);
        default: return launch256_t<0, EPI_NONE, ACM, BCM, 0>(a, s);
    }
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
    if (b.a_cm && b.b_cm) return launch256_t<3, EPI_NONE, true, true, 0>(b, s, splits, 0, -1, 0, per);
    if (b.b_cm) return launch256_t<3, EPI_NONE, false, true, 0>(b, s, splits, 0, -1, 0, per);
    return launch256_t<3, EPI_NONE, true, false, 0>(b, s, splits, 0, -1, 0, per);
}

// forward layout with another DMA schedule (tuning / A-B measurement through vila_gemm_force_sched)
int launch_gemm256_sched(const GemmArgs& a, int sched, hipStream_t s) {
    switch (sched) {
        case 1: return launch256_t<0, EPI_NONE, false, false, 1>(a, s);
        case 2: return launch256_t<0, EPI_NONE, false, false, 2>(a, s);
        case 9: return launch256_t<0, EPI_NONE, false, false, 9>(a, s);
        default: return launch256_t<0, EPI_NONE, false, false, 0>(a, s);
    }
}
