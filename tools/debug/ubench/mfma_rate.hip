// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 in a single dependent chain vs 2/4 independent chains,
// 1..3 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int CHAINS>
__global__ void __launch_bounds__(256) k(float *out, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) acc[c][r] = threadIdx.x * 0.001f + c;
    float a = threadIdx.x * 0.01f, b = 1.0f + threadIdx.x * 0.001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16 / CHAINS; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = MFMA(a, b, acc[c]);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int blocks, int iters, const char *tag) {
    float *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)blocks * 4 * iters * 16 / 1024.0;       // 4 waves/block, 1024 SIMDs
    printf("%s chains=%d blocks=%d: %.1f us, %.1f ns per MFMA per SIMD (=%.1f cycles at 2.4 GHz), %.1f TFLOP/s\n", tag, CHAINS,
           blocks, ms * 1e3, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4,
           (double)blocks * 4 * iters * 16 * 4096 / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main() {
    for (int blocks : {256, 512, 768}) {
        run<1>(blocks, 400, "dep");
        run<2>(blocks, 400, "2ch");
        run<4>(blocks, 400, "4ch");
    }
    return 0;
}
