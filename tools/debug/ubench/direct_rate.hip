// Microbenchmark: A operands straight from global memory (packed in lane order), no LDS staging, no per-stage barrier.
// Per stage per wave: 4 x global_load_dwordx4 (prefetched DIST stages ahead) + 16 dependent MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int WAVES_PER_SIMD>
__global__ void __launch_bounds__(256, WAVES_PER_SIMD) k(const float *w, float *out, int stages) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc, x;
    for (int r = 0; r < 16; r++) { acc[r] = 0.f; x[r] = 0.01f * (r + lane); }
    // stream layout: [stage][wave][q][lane] float4
    const f32x4v *ws = reinterpret_cast<const f32x4v *>(w) + wave * 256 + lane;
    f32x4v a0[4], a1[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { a0[q] = ws[q * 64]; a1[q] = ws[1024 + q * 64]; }
    for (int s = 0; s < stages; s += 2) {
        f32x4v n0[4], n1[4];
        const f32x4v *p = ws + (size_t)((s + 2) & 62) * 1024;
#pragma unroll
        for (int q = 0; q < 4; q++) n0[q] = p[q * 64];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int pp = 0; pp < 4; pp++) acc = MFMA(a0[q][pp], x[4 * q + pp], acc);
#pragma unroll
        for (int q = 0; q < 4; q++) n1[q] = p[1024 + q * 64];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int pp = 0; pp < 4; pp++) acc = MFMA(a1[q][pp], x[4 * q + pp], acc);
#pragma unroll
        for (int q = 0; q < 4; q++) { a0[q] = n0[q]; a1[q] = n1[q]; }
    }
    float sum = 0;
    for (int r = 0; r < 16; r++) sum += acc[r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int WPS>
void run(int blocks, int stages) {
    float *out, *w;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&w, 64 * 4096 * 4);
    hipMemset(w, 0, 64 * 4096 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<WPS>, dim3(blocks), dim3(256), 0, 0, w, out, stages);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WPS>, dim3(blocks), dim3(256), 0, 0, w, out, stages);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)blocks * 4 * stages * 16 / 1024.0;
    printf("direct A operands, %d waves/SIMD bound, blocks=%4d stages=%d: %7.1f us  %.1f ns/MFMA/SIMD\n", WPS, blocks, stages,
           ms * 1e3, ms * 1e6 / mfma_per_simd);
    hipFree(out); hipFree(w);
}
int main() {
    for (int blocks : {256, 512, 768, 1024}) run<4>(blocks, 48);
    for (int blocks : {768, 1536}) run<3>(blocks, 48);
    run<4>(739, 46);
    return 0;
}
