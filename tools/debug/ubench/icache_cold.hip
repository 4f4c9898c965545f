// What does straight-line code cost the FIRST time it runs?  Every launch starts with the instruction caches invalidated, and
// the edge kernels are ~28 KB of straight-line code per instance that a workgroup executes once.  The body here is NI
// independent v_fma_f32 (8 bytes each, 8 rotating registers: no dependency stalls, 4 issue cycles each), emitted NI times in a row
// (no loop) and run TWICE by an outer loop of two: pass 0 finds the code nowhere, pass 1 finds it in the instruction cache.
// A second flavour interleaves one MFMA per NV v_fma (the matrix phases' fetch demand: fewer bytes per cycle).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define F8 "v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n" \
           "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
#define FMA8() asm volatile(F8 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(c));

template <int KB, int MF>
__global__ void __launch_bounds__(256, 2) k_line(float *out) {
    float v[8]; for (int i = 0; i < 8; i++) v[i] = 0.5f + i + threadIdx.x;
    const float c = 0.999f;
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * threadIdx.x + i); b[i] = (_Float16)(0.5f + i); }
    f16v acc; for (int r = 0; r < 16; r++) acc[r] = 0.f;
    unsigned long long t[3];
    t[0] = __builtin_readcyclecounter();
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
        for (int u = 0; u < KB * 16; u++) {         // 16 x 8 instructions x 8 bytes = 1 KB per unit of KB
            FMA8()
            if (MF) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        asm volatile("s_nop 0" ::: "memory");
        t[pass + 1] = __builtin_readcyclecounter();
    }
    float s = 0;
    for (int r = 0; r < 8; r++) s += v[r];
    for (int r = 0; r < 16; r++) s += acc[r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x + 4096] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = (float)(t[1] - t[0]);
        out[2 * w + 1] = (float)(t[2] - t[1]);
    }
}
template <int KB, int MF> void run(float *d, int grid) {
    float *h = new float[4096];
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((k_line<KB, MF>), dim3(grid), dim3(256), 0, 0, d);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, d, 4096 * 4, hipMemcpyDeviceToHost);
    double c0 = 0, c1 = 0, m0 = 0; const int nw = grid * 4;
    for (int w = 0; w < nw && w < 2048; w++) { c0 += h[2 * w]; c1 += h[2 * w + 1]; if (h[2 * w] > m0) m0 = h[2 * w]; }
    const int n = nw < 2048 ? nw : 2048;
    printf("%2d KB of v_fma%s, %3d workgroups of 4 waves: first pass %7.0f cycles (max %7.0f), second pass %7.0f  -> +%5.0f cycles, %.1f per 64-byte line\n",
           KB, MF ? " + 1 MFMA per 8" : "", grid, c0 / n, m0, c1 / n, (c0 - c1) / n, (c0 - c1) / n / (KB * 16.0 * (MF ? 9.0 / 8.0 : 1.0)));
    delete[] h;
}
int main() {
    float *d; (void)hipMalloc(&d, (512 * 256 + 4096) * 4);
    run<8, 0>(d, 256); run<16, 0>(d, 256); run<28, 0>(d, 256); run<28, 0>(d, 512); run<28, 0>(d, 1);
    run<8, 1>(d, 256); run<28, 1>(d, 256); run<28, 1>(d, 512);
    return 0;
}
