// Microbenchmark of the edge-kernel stage structure: per stage each wave does 4 ds_read_b128 (A operand) + 16 dependent
// MFMAs, [optionally 4 global loads + 4 ds_write_b128], one barrier.  Varies workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int MODE, int UNROLL = 1>   // 0: LDS reads + MFMA + barrier; 1: + global load/LDS store of the next chunk; 2: MFMA + barrier only
__global__ void __launch_bounds__(256, 3) k(const float *w, float *out, int stages, int lds_pad_kb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *wbuf0 = smem, *wbuf1 = smem + 128 * 36;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * 36; i += 256) smem[i] = 0.001f * (i & 255);
    __syncthreads();
    f32x16 acc, x;
    for (int r = 0; r < 16; r++) { acc[r] = 0.f; x[r] = 0.01f * (r + lane); }
    int cur = 0;
#pragma unroll UNROLL
    for (int s = 0; s < stages; s++) {
        f32x4v pre[4];
        if (MODE == 1) {
            const float *g = w + (size_t)(s & 31) * 4096;
#pragma unroll
            for (int m = 0; m < 4; m++) pre[m] = *reinterpret_cast<const f32x4v *>(g + 4 * (tid + 256 * m));
        }
        const float *cb = cur ? wbuf1 : wbuf0;
        const float *base = cb + (32 * wave + (lane & 31)) * 36 + 4 * (lane >> 5);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4v a = MODE == 2 ? f32x4v{1.f, 2.f, 3.f, 4.f} : *reinterpret_cast<const f32x4v *>(base + 8 * q);
#pragma unroll
            for (int p = 0; p < 4; p++) acc = MFMA(a[p], x[4 * q + p], acc);
        }
        if (MODE == 1) {
            float *ob = cur ? wbuf0 : wbuf1;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                int idx = tid + 256 * m, row = idx >> 3, c4 = idx & 7;
                *reinterpret_cast<f32x4v *>(ob + row * 36 + 4 * c4) = pre[m];
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    float sum = 0;
    for (int r = 0; r < 16; r++) sum += acc[r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int MODE, int UNROLL = 1>
void run(int blocks, int stages, size_t smem, const char *tag) {
    float *out, *w;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMalloc(&w, 32 * 4096 * 4);
    hipMemset(w, 0, 32 * 4096 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, UNROLL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(blocks), dim3(256), smem, 0, w, out, stages, 0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(blocks), dim3(256), smem, 0, w, out, stages, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)blocks * 4 * stages * 16 / 1024.0;
    printf("%-34s blocks=%4d: %7.1f us  %.1f ns/MFMA/SIMD  per-WG stage %.0f ns\n", tag, blocks, ms * 1e3,
           ms * 1e6 / mfma_per_simd, ms * 1e6 / stages);
    hipFree(out); hipFree(w);
}
int main() {
    const size_t smem = 53248;
    for (int blocks : {256, 512, 768}) {
        run<2>(blocks, 200, smem, "MFMA + barrier");
        run<0>(blocks, 200, smem, "LDS A-reads + MFMA + barrier");
        run<1>(blocks, 200, smem, "+ global load / LDS store of chunk");
    }
    printf("47 stages per workgroup, rolled loop vs straight-line code (instruction fetch):\n");
    for (int blocks : {256, 768, 1536}) {
        run<1, 1>(blocks, 47, smem, "rolled");
        run<1, 47>(blocks, 47, smem, "fully unrolled");
    }
    return 0;
}
