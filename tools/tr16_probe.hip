// Empirically pin the lane/element mapping of ds_read_b64_tr_b16 on gfx950 (the guides describe it in prose only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    // mode 0: every lane passes the address of "its" 8-byte row: row-major [16 rows][4] per 16-lane group
    // mode 1: lane address = base + (l&15)*2 bytes + (l>>4)*128 bytes  (guide formula with per-lane column)
    int byte_off = mode == 0 ? l * 8 : ((l & 15) * 2 + (l >> 4) * 128);
    if (mode == 2) byte_off = (l >> 4) * 128 + ((l & 15) >> 2) * 32 + (l & 3) * 8;   // 4 rows of 16 el, lane picks 4-el chunk
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)((char*)lds + byte_off));
    s16x4 s = __builtin_bit_cast(s16x4, v);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)s[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
