// What does a lone wave's MFMA stage of the edge kernels cost?  12 v_mfma_f32_32x32x16_f16 per stage on two accumulator chains
// (the order of mfma_h<2>), operands rotating over register sets, with the stage fences of the kernel (sched_barrier + empty asm
// on the accumulators), with normal or SUBNORMAL f16 values in the "lo" operands.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
template <int MODE>
__global__ void __launch_bounds__(256, 2) k_stage(float *out, int iters, float lo_scale) {
    h8 a[4], xh[2][2], xl[2][2];
    for (int q = 0; q < 4; q++)
        for (int i = 0; i < 8; i++) a[q][i] = (_Float16)((0.001f * threadIdx.x + i + q) * (q & 1 ? lo_scale : 1.f));
    for (int r = 0; r < 2; r++)
        for (int s = 0; s < 2; s++)
            for (int i = 0; i < 8; i++) {
                xh[r][s][i] = (_Float16)(0.5f + i + r + s);
                xl[r][s][i] = (_Float16)((0.25f + i) * lo_scale);
            }
    f16v acc[2];
    for (int c = 0; c < 2; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; s++) {
#pragma unroll
            for (int r = 0; r < 2; r++) acc[r] = MF(a[2 * s], xh[r][s], acc[r]);
#pragma unroll
            for (int r = 0; r < 2; r++) acc[r] = MF(a[2 * s], xl[r][s], acc[r]);
#pragma unroll
            for (int r = 0; r < 2; r++) acc[r] = MF(a[2 * s + 1], xh[r][s], acc[r]);
        }
        if (MODE >= 1) {
            asm volatile("" ::"v"(acc[0][0]));
            asm volatile("" ::"v"(acc[1][0]));
        }
        if (MODE >= 2) {       // one VALU operation on each accumulator per stage
#pragma unroll
            for (int c = 0; c < 2; c++) acc[c][0] *= 0.5f;
        }
    }
    float s = 0;
    for (int c = 0; c < 2; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (12.f * iters);
}
template <int MODE> void run(float *d, float lo_scale, const char *what) {
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k_stage<MODE>, dim3(256), dim3(256), 0, 0, d, 4000, lo_scale);
    hipDeviceSynchronize();
    float r[1]; hipMemcpy(r, d, 4, hipMemcpyDeviceToHost);
    printf("%-72s %.1f cycles per MFMA\n", what, r[0]);
}
int main() {
    float *d; hipMalloc(&d, (256 * 256 + 2) * 4);
    run<0>(d, 1.f, "two chains, rotating operands, lo operands normal");
    run<0>(d, 1e-6f, "two chains, rotating operands, lo operands SUBNORMAL f16");
    run<0>(d, 0.f, "two chains, rotating operands, lo operands zero");
    run<1>(d, 1.f, "+ stage fences (sched_barrier, asm on the accumulators), normal");
    run<1>(d, 1e-6f, "+ stage fences, SUBNORMAL lo");
    run<2>(d, 1e-6f, "+ one VALU op on each accumulator per stage, SUBNORMAL lo");
    return 0;
}
