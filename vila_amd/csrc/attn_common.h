// Shared pieces of the DMA-ring attention kernels (attn.hip: forward; attn_bwd.hip: dQ and dK / dV).
#pragma once
#include "kernels.h"

#define NEG_BIG (-1.0e30f)
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4v;
typedef __attribute__((address_space(3))) bf16x4v lds_bf16x4;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

template <int HD> struct AttnDma {
    static constexpr int CH = HD / 8;                       // valid 16-B chunks per K / V row
    static constexpr int CHP = (HD == 72) ? 10 : CH;        // chunks per LDS row
    static constexpr int ROWB = CHP * 16;                   // LDS row bytes
    static constexpr int KK = (HD + 31) / 32, DN = (HD + 15) / 16;
    static constexpr int KT = 64;
    static constexpr int IMG = KT * ROWB;                   // one K or V image
    static constexpr int STAGE = 2 * IMG;
    static constexpr int NST = (HD == 128) ? 4 : 6;
    static constexpr int PW = 16 * CHP;                     // DMA slots (16 B) per wave and tile: 2 * 64 * CHP / 8 waves
    static constexpr int P = (PW + 63) / 64;                // DMA instructions per wave and tile (the last one may be partial)
    static constexpr int OSTR = HD + 8;                     // O staging row stride (elements)
    static_assert(HD == 64 || HD == 72 || HD == 128, "head dims of the path");
    // position (16-B slot inside the LDS row) of chunk c of row r; an involution in c for fixed r
    __device__ static __forceinline__ int swz_k(int r, int c) {
        if (HD == 128) return c ^ (r & 15);
        if (HD == 64) return c ^ ((r >> 1) & 7);
        return c;
    }
    __device__ static __forceinline__ int swz_v(int r, int c) {
        if (HD == 128) return (((c >> 1) ^ (r & 7)) << 1) | (c & 1);
        if (HD == 64) return (((c >> 1) ^ ((r >> 1) & 3)) << 1) | (c & 1);
        return c;
    }
};

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt is 6 bits on gfx9");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
// wait until at most `ahead` tiles of P DMA instructions each are still in flight (ahead is block-uniform, 0 .. MAXA)
template <int P, int MAXA> __device__ __forceinline__ void wait_tiles_ahead(int ahead) {
    if constexpr (MAXA == 0) { wait_vmcnt<0>(); }
    else { if (ahead >= MAXA) wait_vmcnt<MAXA * P>(); else wait_tiles_ahead<P, MAXA - 1>(ahead); }
}

// max over lane l and lane l ^ 16 (resp. l ^ 32) without the LDS crossbar: gfx950's row swaps.  v_permlane16_swap exchanges the odd
// 16-lane rows of its first operand with the even rows of the second, v_permlane32_swap the upper half of the first with the lower half
// of the second; fed the same value twice, the two results hold (x[l], x[l ^ 16]) resp. (x[l], x[l ^ 32]) in some order for every lane.
__device__ __forceinline__ float xor16_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

