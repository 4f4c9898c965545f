// What does a dependent kernel launch cost on this GPU?  Chains of tiny kernels on one stream, wall time per kernel:
//   empty        : nothing
//   load1        : every thread reads one dword the PREVIOUS kernel wrote (cross-kernel producer/consumer through L2 / MALL)
//   load_chain2  : two dependent loads (index -> row), like eidx -> pts_j in the edge kernels' prologue
// and the same with 493 x 256 threads (edge launch shape) instead of 47 x 512 (node update shape).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <chrono>
__global__ void k_empty(float *a, const float *b, const int *idx, int n) {}
__global__ void k_load1(float *a, const float *b, const int *idx, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    a[t] = b[t] + 1.f;
}
__global__ void k_chain2(float *a, const float *b, const int *idx, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    a[t] = b[idx[t]] + 1.f;
}
__global__ void k_chain3(float *a, const float *b, const int *idx, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int j = idx[t];
    a[t] = b[(int)b[j] % n] + 1.f;
}
typedef void (*kern_t)(float *, const float *, const int *, int);
int main() {
    const int n = 493 * 256;
    float *a, *b; int *idx;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&idx, n * 4);
    std::vector<int> h(n); for (int i = 0; i < n; i++) h[i] = (i * 7919) % n;
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    hipStream_t s; hipStreamCreate(&s);
    struct { const char *name; kern_t k; } ks[] = {{"empty", k_empty}, {"load1", k_load1}, {"chain2", k_chain2}, {"chain3", k_chain3}};
    int shapes[2][2] = {{47, 512}, {493, 256}};
    for (auto &sh : shapes)
        for (auto &k : ks) {
            const int iters = 2000;
            for (int w = 0; w < 50; w++) { hipLaunchKernelGGL(k.k, dim3(sh[0]), dim3(sh[1]), 0, s, a, b, idx, n); std::swap(a, b); }
            hipStreamSynchronize(s);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, s);
            for (int i = 0; i < iters; i++) { hipLaunchKernelGGL(k.k, dim3(sh[0]), dim3(sh[1]), 0, s, a, b, idx, n); std::swap(a, b); }
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%3d x %3d  %-7s %.2f us per kernel\n", sh[0], sh[1], k.name, ms * 1e3 / iters);
        }
    return 0;
}
