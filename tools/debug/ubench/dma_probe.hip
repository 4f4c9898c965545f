// Probe: where does `global_load_lds_dwordx4 v[..], off offset:X` land in LDS?  (M0 base, instruction offset, lane*16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(const float *src, float *dump) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = -1.f;
    __syncthreads();
    const float *g = src + lane * 4;
    unsigned keep;
    unsigned base = (unsigned)(size_t)(lds + 256);          // LDS byte address of float 256
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(base) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 4096; i += 64) dump[i] = lds[i];
}

int main() {
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; i++) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 8192 * 4);
    hipMalloc(&o, 4096 * 4);
    hipMemcpy(d, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o);
    std::vector<float> r(4096);
    hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 4096; i++) if (r[i] >= 0) { if (first < 0) first = i; last = i; }
    printf("written LDS floats [%d, %d]\n", first, last);
    for (int i : {256, 257, 260, 511, 512, 513, 767, 768}) printf("lds[%d] = %g\n", i, r[i]);
    int mism = 0;
    for (int i = 0; i < 512; i++) if (r[256 + i] != (float)i) mism++;
    printf("linear image of src[0..511] at lds[256..767]: %s (%d mismatches)\n", mism ? "NO" : "YES", mism);
    return 0;
}
