// What does a hand-off inside a small GROUP of workgroups cost?  (A node-update tile split over CL = 4 workgroups that exchange
// their partial FFN outputs once, instead of each computing the whole FFN: DESIGN.md section 4.1.)
//   phase: every workgroup of a group writes its 8 KB partial (16 residues x 128 floats; one 16-byte store per thread, 512 threads),
//   arrives at the GROUP's counter, waits until all CL have arrived, reads the CL - 1 other partials and sums them in a fixed order.
// 192 workgroups = 48 groups of 4 (96 of 2), all resident.  Two placements: group members at consecutive block indices (four different XCDs:
// workgroups go round-robin over the 8 XCDs) and at block indices that differ by 8 (same XCD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int CL>
__global__ void __launch_bounds__(512) k_groups(unsigned *counters, float *buf, int phases, int same_xcd, unsigned long long *cycles) {
    const int b = blockIdx.x, t = threadIdx.x;
    int g, c;
    if (same_xcd) {            // blocks x, x + 8, x + 16, x + 24 of a 32-block window form a group
        const int win = b / (8 * CL), x = b % 8;
        g = win * 8 + x; c = (b / 8) % CL;
    } else { g = b / CL; c = b % CL; }
    unsigned *cnt = counters + g * 32;                     // one 128-byte line per group
    float *mine = buf + ((size_t)g * CL + c) * 2048;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = wall_clock64();
    for (int p = 0; p < phases; p++) {
        // produce: relaxed agent-scope atomic stores (sc1 write-through: no L2 write-back / invalidate instruction anywhere)
#pragma unroll
        for (int q = 0; q < 4; q++)
            __hip_atomic_store(mine + 4 * t + q, (float)(p + b + q) + acc[q] * 1e-30f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);      // stores acknowledged
        __syncthreads();
        if (t == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(p + 1) * (unsigned)CL;
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CL; k++) {                    // fixed order: deterministic sum
            const float *o = buf + ((size_t)g * CL + k) * 2048 + 4 * t;
#pragma unroll
            for (int q = 0; q < 4; q++) acc[q] += __hip_atomic_load(o + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (t == 0 && b == 0) *cycles = wall_clock64() - c0;
    if (acc[0] == 12345.f) buf[0] = acc[1];
}
template <int CL> void run(int same) {
    unsigned *counters; float *buf; unsigned long long *cyc;
    (void)hipMalloc(&counters, 128 * 128); (void)hipMalloc(&buf, 128 * 4 * 2048 * 4); (void)hipMalloc(&cyc, 8);
    const int phases = 2000;          // 192 workgroups (a multiple of 8 CL)
    (void)hipMemset(counters, 0, 128 * 128); (void)hipMemset(buf, 0, 128 * 4 * 2048 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_groups<CL>), dim3(192), dim3(512), 0, 0, counters, buf, phases, same, cyc);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("groups of %d, members on %s: %.2f us per phase (write 8 KB -> arrive -> wait for the group -> read %d partials)\n", CL,
           same ? "the same XCD (block indices 8 apart)" : "different XCDs (consecutive block indices)", ms * 1e3 / phases, CL);
    (void)hipFree(counters); (void)hipFree(buf); (void)hipFree(cyc);
}
int main() {
    run<4>(0); run<4>(1); run<2>(0); run<2>(1);
    return 0;
}
