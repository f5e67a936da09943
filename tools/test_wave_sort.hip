// Stand-alone check of zlib.es_amd/csrc/wave_sort.h on the GPU:
//   hipcc --offload-arch=gfx950 -O3 -I zlib.es_amd/csrc tools/test_wave_sort.hip -o /tmp/test_wave_sort
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "wave_sort.h"

__global__ void k_sort(const uint32_t *in, uint32_t *out) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    out[i] = zgpu_sort::wave_sort64(in[i]);
}

int main() {
    const int waves = 4096, n = waves * 64;
    std::vector<uint32_t> h(n), r(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        const int mode = (i / 64) % 4;
        uint32_t v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
        if (mode == 1) v &= 0xf;                               // many duplicates
        if (mode == 2) v = ((v & 0x1fff) << 6) | (i & 63);     // (hash, lane) keys
        if (mode == 3 && (i & 63) > 40) v = 0xffffffffu;       // padded tail
        h[i] = v;
    }
    uint32_t *di, *dout;
    if (hipMalloc(&di, n * 4) != hipSuccess || hipMalloc(&dout, n * 4) != hipSuccess) return 2;
    if (hipMemcpy(di, h.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    k_sort<<<waves, 64>>>(di, dout);
    if (hipMemcpy(r.data(), dout, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int bad = 0;
    for (int w = 0; w < waves; ++w) {
        std::sort(h.begin() + w * 64, h.begin() + w * 64 + 64);
        for (int l = 0; l < 64; ++l)
            if (h[w * 64 + l] != r[w * 64 + l]) {
                if (bad < 5) printf("wave %d lane %d: got %08x want %08x\n", w, l, r[w * 64 + l], h[w * 64 + l]);
                ++bad;
            }
    }
    printf(bad ? "FAIL (%d mismatches)\n" : "wave_sort64 OK\n", bad);
    return bad ? 1 : 0;
}
