// valu_rates.hip -- issue cost of a few integer instructions on gfx950, relative to v_add_u32 (run on the MI355X box).
// Each kernel runs a dependent chain per lane; 8 waves per SIMD hide the latency, so time ~ issue slots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint64_t *out, uint32_t s) {
    uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ull + blockIdx.x, b = a ^ 0x123456789abcdefull;
    uint32_t x = (uint32_t)a, y = (uint32_t)b, sh = (s + threadIdx.x) & 31u;
#pragma unroll 16
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) { x = x + y; y = y + x; }                                        // 2 x v_add_u32
        if (OP == 1) { a = a << (sh & 63); a ^= b; b = b >> (sh & 63); b ^= a; }      // 2 x 64-bit variable shift (+ 4 xor)
        if (OP == 2) { a = (a << 2) ^ b; b = (b >> 2) ^ a; }                          // 2 x 64-bit constant shift (+ 4 xor)
        if (OP == 3) { x = __builtin_amdgcn_alignbit(x, y, 2) ^ y; y = __builtin_amdgcn_alignbit(y, x, 30) ^ x; }   // 2 alignbit + 2 xor
        if (OP == 4) { x = x * 0x00204081u + y; y = y * 0x01041040u + x; }            // 2 x v_mul_lo_u32 (+ add, maybe mad)
        if (OP == 5) { a = a + b; b = b + a; }                                        // 2 x 64-bit add
        if (OP == 6) { x += (uint32_t)__popcll(a); a = a * 3 + x; }                   // popcll + 64-bit mul-ish
        if (OP == 7) { x += (uint32_t)__builtin_ctzll(a | 1ull << 63); a += x; }      // ctz64
        if (OP == 8) { x = (x << sh) ^ y; y = (y >> sh) ^ x; }                        // 2 x 32-bit variable shift + 2 xor
        if (OP == 9) { x = (x & 0xffffffu) * (y & 0xffffu) + y; y = y ^ x; }          // mul24-able
    }
    out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ x ^ y;
}
template <int OP> void run(const char *name, uint64_t *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, 3u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, 3u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 2048 WGs x 4 waves = 8192 waves over 1024 SIMDs = 8 waves per SIMD; cycles per iteration per wave at 2.4 GHz
    printf("%-40s %8.3f ms  %6.1f cycles/iter/wave (8 waves per SIMD share it)\n", name, ms, ms * 1e-3 * 2.4e9 / ITER / 8);
}
int main() {
    uint64_t *d; hipMalloc(&d, 256 * 8 * 256 * 8);
    run<0>("2 x v_add_u32", d); run<1>("2 x 64-bit variable shift + 2 xor64", d); run<2>("2 x 64-bit const shift + 2 xor64", d);
    run<3>("2 x alignbit + 2 xor", d); run<4>("2 x mul_lo_u32 + add", d); run<5>("2 x add64", d);
    run<6>("popcll + mul64 by 3", d); run<7>("ctz64 + add64", d); run<8>("2 x 32-bit var shift + 2 xor", d); run<9>("mul24 + ops", d);
    return 0;
}
