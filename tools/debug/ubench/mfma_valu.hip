// Can ONE wave hide VALU work behind its own MFMAs?  Loop body = 1 dependent v_mfma_f32_32x32x16_f16 + NV independent v_fma_f32
// (and optionally ND ds_read_b128); one wave per SIMD.  Overlap: ~32 cycles per iteration until NV ~ 7; no overlap: 32 + 4 NV.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NV, int ND>
__global__ void __launch_bounds__(256, 1) k_mix(float *out, int iters) {
    __shared__ f4 lds[1024];
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * threadIdx.x + i); b[i] = (_Float16)(0.5f + i); }
    lds[threadIdx.x] = f4{1.f, 2.f, 3.f, 4.f}; lds[threadIdx.x + 256] = f4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    f16v acc; for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float v[16]; for (int i = 0; i < 16; i++) v[i] = 0.5f + i + threadIdx.x;
    f4 d[4] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i % 16]) : "v"(v[(i + 5) % 16]));
#pragma unroll
            for (int i = 0; i < ND; i++) asm volatile("ds_read_b128 %0, %1" : "=v"(d[i % 4]) : "v"((threadIdx.x & 255) * 16 + 4096 * (i & 1)));
            if (ND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; r++) s += acc[r] + v[r];
    for (int i = 0; i < 4; i++) s += d[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x + 2] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (8.f * iters);
}
template <int NV, int ND> void run(float *d) {
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_mix<NV, ND>), dim3(256), dim3(256), 0, 0, d, 2000);
    (void)hipDeviceSynchronize();
    float r; (void)hipMemcpy(&r, d, 4, hipMemcpyDeviceToHost);
    printf("1 MFMA + %2d v_fma + %d ds_read_b128: %.1f cycles per iteration\n", NV, ND, r);
}
int main() {
    float *d; (void)hipMalloc(&d, (256 * 256 + 2) * 4);
    run<0, 0>(d); run<2, 0>(d); run<4, 0>(d); run<6, 0>(d); run<8, 0>(d); run<12, 0>(d); run<16, 0>(d);
    run<0, 1>(d); run<0, 2>(d); run<4, 1>(d); run<4, 2>(d);
    return 0;
}
