// Probe v_mfma_f32_32x32x16_f16 on gfx950: operand layout, denormal handling, issue rate, and the accuracy of the
// two-way f16 split (hi*hi + hi*lo + lo*hi) against an fp64 reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A: [32 m][16 k] row-major floats (exact f16 values), B: [16 k][32 n]; D: [32][32]
__global__ void k_layout(const float *A, const float *B, float *D) {
    const int l = threadIdx.x, r32 = l & 31, g = l >> 5;
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)A[r32 * 16 + 8 * g + i]; b[i] = (_Float16)B[(8 * g + i) * 32 + r32]; }
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; r++) D[(8 * (r >> 2) + 4 * g + (r & 3)) * 32 + r32] = acc[r];
}

// split product: X [32 m][K] fp32, W [K][32 n] fp32 -> D via 3 MFMAs per 16-deep step
__global__ void k_split(const float *X, const float *W, float *D, int K) {
    const int l = threadIdx.x, r32 = l & 31, g = l >> 5;
    f16v acc = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        h8 ah, al, bh, bl;
        for (int i = 0; i < 8; i++) {
            float xa = X[r32 * K + k0 + 8 * g + i], xb = W[(k0 + 8 * g + i) * 32 + r32];
            _Float16 h1 = (_Float16)xa; ah[i] = h1; al[i] = (_Float16)(xa - (float)h1);
            _Float16 h2 = (_Float16)xb; bh[i] = h2; bl[i] = (_Float16)(xb - (float)h2);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; r++) D[(8 * (r >> 2) + 4 * g + (r & 3)) * 32 + r32] = acc[r];
}

__global__ void k_rate(float *out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.001f * threadIdx.x + i); b[i] = (_Float16)(0.5f + i); }
    f16v acc = {0};
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0; for (int r = 0; r < 16; r++) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (16.f * iters);
}

int main() {
    std::mt19937 rng(1);
    std::vector<float> A(32 * 16), B(16 * 32), D(1024), Dref(1024);
    for (auto &v : A) v = (float)((int)(rng() % 17) - 8);
    for (auto &v : B) v = (float)((int)(rng() % 17) - 8);
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) { float s = 0; for (int k = 0; k < 16; k++) s += A[m * 16 + k] * B[k * 32 + n]; if (s != D[m * 32 + n]) bad++; }
    printf("layout check (lane = row/col, k = 8*(lane>>5)+i; D[m = 8(r>>2)+4(lane>>5)+(r&3)][n = lane&31]): %d mismatches\n", bad);

    // denormals: A = 2^-20 (f16 subnormal), B = 1024 -> expect 16 * 2^-10
    for (auto &v : A) v = ldexpf(1.f, -20);
    for (auto &v : B) v = 1024.f;
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    printf("subnormal f16 input 2^-20 x 1024 summed over k=16: got %g, exact %g  (0 => inputs flushed)\n", D[0], 16 * ldexp(1.0, -10));

    // split accuracy, K = 128
    const int K = 128;
    std::vector<float> X(32 * K), W(K * 32);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto &v : X) v = nd(rng) * 3.f;
    for (auto &v : W) v = nd(rng) * 0.1f;
    float *dX, *dW;
    hipMalloc(&dX, X.size() * 4); hipMalloc(&dW, W.size() * 4);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split, dim3(1), dim3(64), 0, 0, dX, dW, dD, K);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double emax = 0, e32max = 0, ref_rms = 0;
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) {
        double s = 0; float s32 = 0;
        for (int k = 0; k < K; k++) { s += (double)X[m * K + k] * W[k * 32 + n]; s32 = fmaf(X[m * K + k], W[k * 32 + n], s32); }
        emax = fmax(emax, fabs(D[m * 32 + n] - s)); e32max = fmax(e32max, fabs(s32 - s)); ref_rms += s * s;
    }
    printf("K=128 split-f16 (3 MFMA): max abs err %.3e ; sequential fp32 fma: %.3e ; rms of result %.3f\n", emax, e32max, sqrt(ref_rms / 1024));

    float *dout; hipMalloc(&dout, 256 * 4 * 256 * 4);
    hipLaunchKernelGGL(k_rate, dim3(256 * 4), dim3(64), 0, 0, dout, 2000);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_rate, dim3(256 * 4), dim3(64), 0, 0, dout, 2000);
    float cyc; hipMemcpy(&cyc, dout, 4, hipMemcpyDeviceToHost);
    printf("dependent v_mfma_f32_32x32x16_f16 chain, one wave per SIMD: %.1f clock64 ticks per MFMA (100 MHz ticks => x%.1f ns)\n", cyc, 10.0);
    return 0;
}
