// hbm_rates.hip -- what pure write / read / copy streams reach on this chip (run on the MI355X box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(1024) void k_fill(uint4 *p, size_t n16) {          // every workgroup owns contiguous 32 KiB pieces
    const size_t per_wg = 2048;                                                // uint4 per piece
    for (size_t base = (size_t)blockIdx.x * per_wg; base < n16; base += (size_t)gridDim.x * per_wg)
        for (size_t i = threadIdx.x; i < per_wg && base + i < n16; i += 1024) p[base + i] = make_uint4(0, 0, 0, 0);
}
__global__ __launch_bounds__(1024) void k_fill_nt(uint4 *p, size_t n16) {
    const size_t per_wg = 2048;
    for (size_t base = (size_t)blockIdx.x * per_wg; base < n16; base += (size_t)gridDim.x * per_wg)
        for (size_t i = threadIdx.x; i < per_wg && base + i < n16; i += 1024) {
            typedef uint32_t V4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(V4{0u, 0u, 0u, 0u}, reinterpret_cast<V4 *>(&p[base + i]));
        }
}
__global__ __launch_bounds__(1024) void k_read(const uint4 *p, size_t n16, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 1024) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t bytes = 16ull << 30, n16 = bytes / 16;
    uint4 *p; uint32_t *o; hipMalloc(&p, bytes); hipMalloc(&o, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto &&f, double gb) {
        f(); hipDeviceSynchronize();
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, ms, gb / ms);
    };
    time("hipMemsetAsync 16 GiB", [&] { hipMemsetAsync(p, 0, bytes, 0); }, bytes / 1e9);
    for (int g : {512, 2048, 16384, 524288}) {
        char nm[64]; snprintf(nm, 64, "fill kernel, %d workgroups", g);
        time(nm, [&] { hipLaunchKernelGGL(k_fill, dim3(g), dim3(1024), 0, 0, p, n16); }, bytes / 1e9);
    }
    time("fill kernel, nontemporal, 2048 workgroups", [&] { hipLaunchKernelGGL(k_fill_nt, dim3(2048), dim3(1024), 0, 0, p, n16); }, bytes / 1e9);
    time("read kernel, 4096 workgroups", [&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(1024), 0, 0, p, n16, o); }, bytes / 1e9);
    return 0;
}
