// Shared device/host helpers for the vila_hip library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <type_traits>

typedef uint16_t bf16_t;  // raw bfloat16 bits; all activations / weights are bf16 in HBM

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define WAVE 64

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even, NaN kept quiet
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// block reduce (256 threads); scratch = 4 floats of LDS
__device__ __forceinline__ float block_sum_256(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // 0.5 x (1 + tanh(u)) = x * sigmoid(2u) = x / (1 + e^(-2u)): one v_exp_f32 and one v_rcp_f32 instead of the library tanhf (a branchy
    // ~30-instruction routine that made the fc1 epilogue of the tower cost 6 us of a 31-us GEMM); relative error ~1e-6, the result is rounded to bf16
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u2 = 2.f * k0 * (x + k1 * x * x * x);
    return x * __builtin_amdgcn_rcpf(1.f + __expf(-u2));
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

// bijective XCD-aware remap of a 1-D grid: blocks that share an XCD (bid % 8) get a contiguous id range
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---- LDS transpose read as inline asm ---------------------------------------------------------------------------------------------
// `__builtin_amdgcn_ds_read_tr16_b64_*` is modelled by the compiler as an LDS access that may also WRITE, so behind any LDS-DMA
// (`global_load_lds`) still in flight it gets a conservative `s_waitcnt vmcnt(0)` in front of it — which drains the whole DMA ring
// once per phase (found in round 3 in the contraction-major GEMMs and in the attention kernels).  The asm form is invisible to that
// pass; the price is that its result is unprotected until OUR wait: retire it with `lds_wait<N>(regs...)`, which ties the destination
// registers to the `s_waitcnt lgkmcnt(N)` so that no consumer can be scheduled in front of it.  LDS operations return in order, so
// compiler-tracked reads in between only make either side's counted waits more conservative, never wrong.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF> __device__ __forceinline__ u32x2 ds_read_tr16_b64(uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field is 16 bits");
    u32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int N> __device__ __forceinline__ void lds_wait_imm() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N));
}
template <int N, typename... R> __device__ __forceinline__ void lds_wait(R&... regs) {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
    if constexpr (sizeof...(R) == 2) {
        auto tie = [](auto& a, auto& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); };
        tie(regs...);
    } else if constexpr (sizeof...(R) == 4) {
        auto tie = [](auto& a, auto& b, auto& c, auto& d) { asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N)); };
        tie(regs...);
    } else if constexpr (sizeof...(R) == 8) {
        auto tie = [](auto& a, auto& b, auto& c, auto& d, auto& e, auto& f, auto& g, auto& h) {
            asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N)); };
        tie(regs...);
    } else {
        static_assert(sizeof...(R) == 2, "lds_wait: 2, 4 or 8 registers");
    }
}
// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); static_for<B + 1, E>(f); }
}

// ---- host side ----
void vila_set_error(const char* fmt, ...);
#define VILA_FAIL(code, ...) do { vila_set_error(__VA_ARGS__); return (code); } while (0)
#define VILA_REQUIRE(cond, ...) do { if (!(cond)) { vila_set_error(__VA_ARGS__); return -1; } } while (0)
#define VILA_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    vila_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return -2; } } while (0)
#define VILA_LAUNCH_CHECK() VILA_HIP(hipGetLastError())
#define VILA_TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
