// Microbenchmark: cost of random LDS reads by width and address alignment on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned_bench.hip -o tools/lds_unaligned_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((aligned(1))) v4u_u;
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef v2u __attribute__((aligned(1))) v2u_u;
typedef uint32_t __attribute__((aligned(1))) u32_u;

template <int WIDTH>
__global__ void __launch_bounds__(1024) k(uint32_t *out, unsigned long long *cyc, uint32_t amask, int iters) {
    __shared__ __align__(16) uint8_t lds[65536 + 64];
    for (int i = threadIdx.x; i < 65536 / 4; i += 1024) ((uint32_t *)lds)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 747796405u + 12345u, acc = 0;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        const uint32_t a = ((x >> 8) & 0xffffu) & amask;
        if (WIDTH == 16) { v4u v = *(const v4u_u *)(lds + a); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        if (WIDTH == 8) { v2u v = *(const v2u_u *)(lds + a); acc ^= v.x ^ v.y; }
        if (WIDTH == 4) { acc ^= *(const u32_u *)(lds + a); }
        x ^= acc & 1;  // dependent chain: the next address needs this read
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    uint32_t *out; unsigned long long *cyc, h[256];
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    const uint32_t masks[4] = {0xfff0u, 0xfff8u, 0xfffcu, 0xffffu};
    const char *names[4] = {"16B-aligned", "8B-aligned", "4B-aligned", "byte"};
    for (int w = 0; w < 3; ++w)
        for (int m = 0; m < 4; ++m) {
            if (w == 0) hipLaunchKernelGGL(k<16>, dim3(256), dim3(1024), 0, 0, out, cyc, masks[m], iters);
            if (w == 1) hipLaunchKernelGGL(k<8>, dim3(256), dim3(1024), 0, 0, out, cyc, masks[m], iters);
            if (w == 2) hipLaunchKernelGGL(k<4>, dim3(256), dim3(1024), 0, 0, out, cyc, masks[m], iters);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
            // 16 waves per CU issue one read each per iteration: cycles per wave-instruction at the LDS
            printf("width %2d %-12s: %.1f cycles/iteration/wave -> %.1f LDS cycles per wave-instruction (16 waves)\n",
                   w == 0 ? 16 : w == 1 ? 8 : 4, names[m], s / 256 / iters, s / 256 / iters / 16);
        }
    return 0;
}
