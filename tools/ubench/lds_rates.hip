// lds_rates.hip -- throughput of LDS operations with scattered addresses on gfx950 (run on the MI355X box):
// what one record's trip through an LDS counting sort costs.  Every lane issues ITER operations on pseudo-random
// dword addresses among NADDR entries (the digit counters of a sort tile: 128 / 256; or 16 K: its parking area);
// 16 waves per CU as in the sort kernels (2 workgroups of 512).  Reported: wave-instructions per microsecond per CU and
// lane-operations per cycle per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
// OP 0: ds_add (no return)  1: ds_add_rtn (result consumed at the end)  2: ds_read  3: ds_write  4: ds_add_rtn -> dependent ds_write (parking)
// 5: sequential ds_read (lane-consecutive)   6: ds_add_rtn, 8 in flight before the results are used
template <int OP, int NADDR>
__global__ __launch_bounds__(512) void k(uint32_t *out, uint32_t seed) {
    __shared__ uint32_t cnt[NADDR];
    __shared__ uint32_t park[16384 + 64];
    for (int i = threadIdx.x; i < NADDR; i += 512) cnt[i] = 0;
    for (int i = threadIdx.x; i < 16384 + 64; i += 512) park[i] = i;
    __syncthreads();
    uint32_t x = (threadIdx.x + blockIdx.x * 512u) * 2654435761u + seed, acc = 0;
#pragma unroll 8
    for (int i = 0; i < ITER; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t d = (x >> 9) & (NADDR - 1);
        if (OP == 0) __hip_atomic_fetch_add(&cnt[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 1 || OP == 6) acc += atomicAdd(&cnt[d], 1u);
        if (OP == 2) acc += cnt[d];
        if (OP == 3) park[(x >> 9) & 16383u] = x;
        if (OP == 4) { const uint32_t p = atomicAdd(&cnt[d], 1u); park[(p + d * 64u) & 16383u] = x; }
        if (OP == 5) acc += park[(threadIdx.x + i * 512u) & 16383u];
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc + cnt[threadIdx.x & (NADDR - 1)] + park[threadIdx.x];
}
template <int OP, int NADDR> void run(const char *name, uint32_t *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 2 * 4;                            // 4 rounds of two workgroups per CU
    hipLaunchKernelGGL((k<OP, NADDR>), dim3(grid), dim3(512), 0, 0, d, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, NADDR>), dim3(grid), dim3(512), 0, 0, d, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)grid * 8 * ITER, per_cu = wave_instr / 256.0;
    printf("%-44s %4d addr %8.3f ms  %7.1f wave-instr/us/CU  %5.2f lane-ops/cycle/CU (2.2 GHz)\n", name, NADDR, ms, per_cu / (ms * 1e3), per_cu * 64 / (ms * 1e-3 * 2.2e9));
}
int main() {
    uint32_t *d; hipMalloc(&d, 256 * 8 * 512 * 4);
    run<0, 128>("ds_add (no return), random", d); run<0, 256>("ds_add (no return), random", d);
    run<1, 128>("ds_add_rtn, random", d); run<1, 256>("ds_add_rtn, random", d);
    run<2, 128>("ds_read, random", d); run<2, 256>("ds_read, random", d);
    run<3, 128>("ds_write, random over 16 K dwords", d);
    run<4, 128>("ds_add_rtn -> ds_write (parking chain)", d); run<4, 256>("ds_add_rtn -> ds_write (parking chain)", d);
    run<5, 128>("ds_read, lane-consecutive", d);
    return 0;
}
