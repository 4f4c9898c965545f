// Micro-repro for the "occupancy hazard" of gfx950's v_cvt_pk_f16_f32 (pp_edge_f16.hip cvt2: with that instruction the split-f16
// kernels were only correct with one wave per SIMD; lanes 16-31 and 48-63 of some tiles came out wrong).  Hypothesis tested here:
// the packed conversion runs at 16 lanes per pass, and a dependent instruction issued right behind it reads each 32-lane half
// when only its first 16 lanes have been written -- a missing wait state that neither the hardware interlock nor the compiler
// covers.  Every lane converts a pair (x0, x1) and feeds the packed result to a consumer, once through a SUSPECT sequence (the
// consumer N wait states behind the producer, N = 0, 1, 2: the assembler adds nothing inside one asm block) and once through a
// SAFE one (two `s_nop 7` in between); any lane whose two results differ is counted per 16-lane group.
//   producers: v_cvt_pk_f16_f32 (gfx950, round to nearest even), v_cvt_pkrtz_f16_f32 (what the kernels ship)
//   consumers: v_fma_mix_f32 (the split's residual), v_pk_max_u16 ... wait-free VALU; ds_write_b32 + ds_read_b32 of a neighbour's word
//   occupancy: 1, 2, 4 waves per SIMD; with and without MFMAs between the iterations (the real kernels' neighbours)
//     hipcc -O3 --offload-arch=gfx950 -o /tmp/cvt_pk_hazard tools/debug/ubench/cvt_pk_hazard.hip && /tmp/cvt_pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define PRODUCER_PK "v_cvt_pk_f16_f32 %0, %2, %3\n"
#define PRODUCER_RTZ "v_cvt_pkrtz_f16_f32 %0, %2, %3\n"
#define CONSUMER_MIX "v_fma_mix_f32 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n"
#define SEQ(P, GAP, C) asm volatile(P GAP C : "=&v"(hp), "=&v"(d) : "v"(x0), "v"(x1))

// PROD 2: the packed conversion with an INLINE CONSTANT as its second source (what hipcc emits for the geometry operands' {distance, 0}
// pair: the one call site at which -DPP_X_CVT_PK_SITES bisected the failure of the real kernels)
#define PRODUCER_PK0 "v_cvt_pk_f16_f32 %0, %2, 0\n"
template <int PROD, int GAP>
__device__ __forceinline__ void conv_mix(float x0, float x1, unsigned &hp, float &d) {
    if constexpr (PROD == 2) {
        if constexpr (GAP == 0) SEQ(PRODUCER_PK0, "", CONSUMER_MIX);
        else if constexpr (GAP == 1) SEQ(PRODUCER_PK0, "s_nop 0\n", CONSUMER_MIX);
        else if constexpr (GAP == 2) SEQ(PRODUCER_PK0, "s_nop 1\n", CONSUMER_MIX);
        else SEQ(PRODUCER_PK0, "s_nop 7\ns_nop 7\n", CONSUMER_MIX);
    } else if constexpr (PROD == 0) {
        if constexpr (GAP == 0) SEQ(PRODUCER_PK, "", CONSUMER_MIX);
        else if constexpr (GAP == 1) SEQ(PRODUCER_PK, "s_nop 0\n", CONSUMER_MIX);
        else if constexpr (GAP == 2) SEQ(PRODUCER_PK, "s_nop 1\n", CONSUMER_MIX);
        else SEQ(PRODUCER_PK, "s_nop 7\ns_nop 7\n", CONSUMER_MIX);
    } else {
        if constexpr (GAP == 0) SEQ(PRODUCER_RTZ, "", CONSUMER_MIX);
        else if constexpr (GAP == 1) SEQ(PRODUCER_RTZ, "s_nop 0\n", CONSUMER_MIX);
        else if constexpr (GAP == 2) SEQ(PRODUCER_RTZ, "s_nop 1\n", CONSUMER_MIX);
        else SEQ(PRODUCER_RTZ, "s_nop 7\ns_nop 7\n", CONSUMER_MIX);
    }
}
// producer -> ds_write_b32 of the packed word -> barrier-free read-back of the SAME lane's word (LDS ops of a wave are in order)
template <int PROD, int GAP>
__device__ __forceinline__ void conv_lds(float x0, float x1, unsigned &hp, float &d, unsigned lds_addr) {
    unsigned back;
    if constexpr (PROD == 0) {
        if constexpr (GAP == 0) asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\nds_write_b32 %4, %0\nds_read_b32 %1, %4\ns_waitcnt lgkmcnt(0)\n" : "=&v"(hp), "=&v"(back) : "v"(x0), "v"(x1), "v"(lds_addr) : "memory");
        else asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\ns_nop 7\ns_nop 7\nds_write_b32 %4, %0\nds_read_b32 %1, %4\ns_waitcnt lgkmcnt(0)\n" : "=&v"(hp), "=&v"(back) : "v"(x0), "v"(x1), "v"(lds_addr) : "memory");
    } else {
        if constexpr (GAP == 0) asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\nds_write_b32 %4, %0\nds_read_b32 %1, %4\ns_waitcnt lgkmcnt(0)\n" : "=&v"(hp), "=&v"(back) : "v"(x0), "v"(x1), "v"(lds_addr) : "memory");
        else asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\ns_nop 7\ns_nop 7\nds_write_b32 %4, %0\nds_read_b32 %1, %4\ns_waitcnt lgkmcnt(0)\n" : "=&v"(hp), "=&v"(back) : "v"(x0), "v"(x1), "v"(lds_addr) : "memory");
    }
    d = __uint_as_float(back);
}

// CONS 0: v_fma_mix_f32, 1: LDS round trip.  NOISE: MFMAs between iterations.  The compiler-visible form (what -DPP_X_CVT_PK builds:
// __builtin_convertvector + a C-level consumer) is CONS 2: whatever hipcc schedules, including its own hazard handling.
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
template <int PROD, int GAP, int CONS, bool NOISE, int WPS>
__global__ void __launch_bounds__(256, WPS) k(unsigned *bad, int iters, float seed) {
    __shared__ unsigned lds[256];
    __shared__ __attribute__((aligned(16))) unsigned lds16[256 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    float x0 = seed + 0.37f * lane + 0.001f * blockIdx.x, x1 = -seed * 1.7f + 0.11f * lane;
    f32x16 acc;
    h8 a, b;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.01f * (i + lane)); b[i] = (_Float16)(0.02f * i); }
    unsigned nbad = 0;
    const unsigned addr = (unsigned)(size_t)(&lds[tid]);
    for (int it = 0; it < iters; it++) {
        unsigned hs, hr;
        float ds_, dr;
        if constexpr (CONS == 0) { conv_mix<PROD, GAP>(x0, x1, hs, ds_); conv_mix<PROD, 3>(x0, x1, hr, dr); }
        else if constexpr (CONS == 1) { conv_lds<PROD, GAP>(x0, x1, hs, ds_, addr); conv_lds<PROD, 3>(x0, x1, hr, dr, addr); }
        else if constexpr (CONS == 2) {
            const f32x2v xv = {x0, x1};
            const h2v hh = PROD == 0 ? __builtin_convertvector(xv, h2v)
                                     : __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(x0, x1));
            hs = __builtin_bit_cast(unsigned, hh);
            ds_ = x0 - (float)hh[0];
            conv_mix<PROD, 3>(x0, x1, hr, dr);
        }
        if constexpr (CONS == 3 || CONS == 4) {
            // eight values -> four packed words -> one MFMA B operand (CONS 3) / one ds_write_b128 + read-back (CONS 4), the consumer
            // right behind the conversions (compiler-scheduled: its own hazard table decides the wait states); the reference path
            // parks the operand behind sixteen wait states first
            float xs[8];
#pragma unroll
            for (int i = 0; i < 8; i++) xs[i] = (i & 1 ? x1 : x0) * (1.f + 0.03125f * i);
            h8 opd;
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const f32x2v xv = {xs[i], xs[i + 1]};
                const h2v hh = PROD == 0 ? __builtin_convertvector(xv, h2v) : __builtin_bit_cast(h2v, __builtin_amdgcn_cvt_pkrtz(xs[i], xs[i + 1]));
                opd[i] = hh[0]; opd[i + 1] = hh[1];
            }
            h8 safe;
            if constexpr (CONS == 3) {
                f32x16 z;
                for (int i = 0; i < 16; i++) z[i] = 0.f;
                __builtin_amdgcn_sched_barrier(0);
                const f32x16 r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, opd, z, 0, 0, 0);      // right behind the conversions
                __builtin_amdgcn_sched_barrier(0);
                safe = opd;
                asm volatile("s_nop 7\ns_nop 7" : "+v"(safe));
                const f32x16 r2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, safe, z, 0, 0, 0);
                unsigned diff = 0;
                for (int i = 0; i < 16; i++) diff |= __float_as_uint(r1[i]) ^ __float_as_uint(r2[i]);
                hs = diff; hr = 0; ds_ = dr = 0.f;
            } else {
                h8 *slot = reinterpret_cast<h8 *>(lds16) + tid;
                __builtin_amdgcn_sched_barrier(0);
                *slot = opd;                                                                        // right behind the conversions
                __builtin_amdgcn_sched_barrier(0);
                safe = opd;
                asm volatile("s_nop 7\ns_nop 7" : "+v"(safe));
                const h8 back = *reinterpret_cast<volatile h8 *>(slot);
                unsigned diff = 0;
                for (int i = 0; i < 8; i++) diff |= (unsigned)(__builtin_bit_cast(unsigned short, back[i]) ^ __builtin_bit_cast(unsigned short, safe[i]));
                hs = diff; hr = 0; ds_ = dr = 0.f;
            }
        }
        nbad += (hs != hr) || (__float_as_uint(ds_) != __float_as_uint(dr));
        if constexpr (NOISE) {
#pragma unroll
            for (int m = 0; m < 4; m++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        x0 = x0 * 1.00013f + 0.5f;
        x1 = x1 * 0.99991f - 0.25f;
        if (x0 > 6.0e4f) x0 = seed;
        if (x1 < -6.0e4f) x1 = -seed;
    }
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += acc[i];
    if (s == 12345.678f) nbad += 1u << 30;            // keeps the MFMAs alive
    if (nbad) atomicAdd(&bad[(lane >> 4) & 3], nbad);
}

// ---- second suspect: a VALU write to a DATA register of a ds_write_b128 that is still waiting for the LDS -------------------------------
// The bisection of the real kernels (tools/debug/cvt_pk_bisect.sh) ends in geometry_put: with v_cvt_pk_f16_f32 at all four point
// pairs hipcc allocates registers so that `ds_write_b128 v[0:3]; ds_write_b128 v[132:135]; ds_write_b16 ..; v_cvt_f16_f32 v0, ..`
// follow each other -- a VALU instruction overwrites v0 two instructions behind the 16-byte store that reads it.  The hazard table
// asks for ONE wait state there (a store of more than 8 bytes followed by a write to its data registers); the data of a DS store
// moves to the LDS at 2 cycles per dword AFTER the LDS has accepted the instruction, and with other waves' LDS traffic in the queue
// that can be later.  Here: the store pair, a 2-byte store, then `v_mov_b32` into the first data register GAP wait states later;
// the LDS word is read back and compared with what the register held when the store was issued.
template <int GAP, bool NOISE, int WPS>
__global__ void __launch_bounds__(256, WPS) k_war(unsigned *bad, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned slots[3 * 256 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned addr = (unsigned)(size_t)(&slots[tid * 4]);
    f32x16 acc;
    h8 a, b;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(0.01f * (i + lane)); b[i] = (_Float16)(0.02f * i); }
    unsigned nbad = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned val = 0x1000u * it + tid, junk = ~val;
        unsigned back;
#define WAR_SEQ(NOPS)                                                                                                        \
        asm volatile("v_mov_b32 v40, %1\nv_add_u32 v41, 1, %1\nv_add_u32 v42, 2, %1\nv_add_u32 v43, 3, %1\n"                     \
                     "v_mov_b32 v44, %1\nv_mov_b32 v45, %1\nv_mov_b32 v46, %1\nv_mov_b32 v47, %1\n"                              \
                     "s_nop 4\n"                                                                                            \
                     "ds_write_b128 %2, v[40:43]\n"                                                                         \
                     "ds_write_b128 %2, v[44:47] offset:4096\n"                                                             \
                     "ds_write_b16 %2, v44 offset:8192\n" NOPS                                                              \
                     "v_mov_b32 v40, %3\n"                                                                                  \
                     "s_waitcnt lgkmcnt(0)\n"                                                                               \
                     "ds_read_b32 %0, %2\n"                                                                                 \
                     "s_waitcnt lgkmcnt(0)\n"                                                                               \
                     : "=&v"(back) : "v"(val), "v"(addr), "v"(junk)                                                         \
                     : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory")
        if constexpr (GAP == 0) WAR_SEQ("");
        else if constexpr (GAP == 1) WAR_SEQ("s_nop 0\n");
        else if constexpr (GAP == 2) WAR_SEQ("s_nop 3\n");
        else WAR_SEQ("s_nop 7\ns_nop 7\n");
        nbad += back != val;
        if constexpr (NOISE) {
#pragma unroll
            for (int m = 0; m < 2; m++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
    }
    float sacc = 0.f;
    for (int i = 0; i < 16; i++) sacc += acc[i];
    if (sacc == 12345.678f) nbad += 1u << 30;
    if (nbad) atomicAdd(&bad[(lane >> 4) & 3], nbad);
}
template <int GAP, bool NOISE, int WPS>
static void run_war(unsigned *bad) {
    hipMemset(bad, 0, 16);
    hipLaunchKernelGGL((k_war<GAP, NOISE, WPS>), dim3(256 * WPS * 2), dim3(256), 0, 0, bad, 20000);
    unsigned h[4];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("ds_write_b128 data overwritten %d wait states behind the stores  %s  %d waves/SIMD: stale/new words by 16-lane group [%u %u %u %u]%s\n",
           GAP == 0 ? 0 : GAP == 1 ? 1 : GAP == 2 ? 4 : 16, NOISE ? "mfma " : "quiet", WPS, h[0], h[1], h[2], h[3], (h[0] | h[1] | h[2] | h[3]) ? "   <-- HAZARD" : "");
}
template <bool NOISE, int WPS>
static void war_gaps(unsigned *bad) { run_war<0, NOISE, WPS>(bad); run_war<1, NOISE, WPS>(bad); run_war<2, NOISE, WPS>(bad); run_war<3, NOISE, WPS>(bad); }

template <int PROD, int GAP, int CONS, bool NOISE, int WPS>
static void run(unsigned *bad, const char *what) {
    hipMemset(bad, 0, 16);
    const int blocks = 256 * WPS * 2;           // two rounds of WPS workgroups per CU
    hipLaunchKernelGGL((k<PROD, GAP, CONS, NOISE, WPS>), dim3(blocks), dim3(256), 0, 0, bad, 4000, 1.25f);
    unsigned h[4];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    printf("%-18s gap %d  %-9s %s  %d waves/SIMD: mismatching lane-iterations by 16-lane group [%u %u %u %u]%s\n",
           PROD == 0 ? "v_cvt_pk_f16_f32" : PROD == 2 ? "v_cvt_pk x, 0" : "v_cvt_pkrtz", GAP, CONS == 0 ? "fma_mix" : CONS == 1 ? "lds" : CONS == 2 ? "compiler" : CONS == 3 ? "mfma B" : "ds_write128", NOISE ? "mfma " : "quiet",
           WPS, h[0], h[1], h[2], h[3], (h[0] | h[1] | h[2] | h[3]) ? "   <-- HAZARD" : "");
    (void)what;
}
template <int PROD, int CONS, bool NOISE, int WPS>
static void gaps(unsigned *bad) {
    run<PROD, 0, CONS, NOISE, WPS>(bad, "");
    if (CONS == 0) { run<PROD, 1, CONS, NOISE, WPS>(bad, ""); run<PROD, 2, CONS, NOISE, WPS>(bad, ""); }
}
template <int PROD, int CONS>
static void occ(unsigned *bad) {
    gaps<PROD, CONS, false, 1>(bad); gaps<PROD, CONS, false, 2>(bad); gaps<PROD, CONS, false, 4>(bad);
    gaps<PROD, CONS, true, 1>(bad); gaps<PROD, CONS, true, 2>(bad); gaps<PROD, CONS, true, 4>(bad);
}
int main() {
    unsigned *bad;
    hipMalloc(&bad, 16);
    occ<0, 0>(bad); occ<0, 1>(bad); occ<0, 2>(bad); occ<0, 3>(bad); occ<0, 4>(bad);
    occ<1, 0>(bad); occ<1, 1>(bad); occ<1, 3>(bad); occ<1, 4>(bad);
    occ<2, 0>(bad);
    war_gaps<false, 1>(bad); war_gaps<false, 2>(bad); war_gaps<false, 3>(bad); war_gaps<true, 2>(bad); war_gaps<true, 3>(bad);
    hipFree(bad);
    return 0;
}
