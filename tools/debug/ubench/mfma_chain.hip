// Issue interval of v_mfma_f32_32x32x16_f16 when consecutive MFMAs chain on the SAME accumulator (1 chain) or rotate over
// 2, 3, 4 independent accumulators; one wave per SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NCH>
__global__ void __launch_bounds__(256, 1) k_chain(float *out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * threadIdx.x + i); b[i] = (_Float16)(0.5f + i); }
    f16v acc[NCH];
    for (int c = 0; c < NCH; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 12; u++)
#pragma unroll
            for (int c = 0; c < NCH; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < NCH; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = (float)(t1 - t0) / (12.f * NCH * iters);
        out[1] = (float)(w1 - w0) * 10.f / (12.f * NCH * iters);     // wall clock: 100 MHz
    }
}
template <int NCH> void run(float *d) {
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_chain<NCH>, dim3(256), dim3(256), 0, 0, d, 2000);
    hipDeviceSynchronize();
    float r[2]; hipMemcpy(r, d, 8, hipMemcpyDeviceToHost);
    printf("%d chain(s): %.1f core cycles, %.2f ns per MFMA per SIMD\n", NCH, r[0], r[1]);
}
int main() {
    float *d; hipMalloc(&d, (256 * 256 + 2) * 4);
    run<1>(d); run<2>(d); run<3>(d); run<4>(d);
    return 0;
}
