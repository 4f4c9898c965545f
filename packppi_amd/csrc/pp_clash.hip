// Torsion angles -> atom14 coordinates, clash loss with analytic chi gradient, and the proximal
// Adam loop.  Reference: components/__init__.py:76-120, features.py:95-194 (reconstruction);
// clash.py:7-99,102-254,335-365 (loss); optimize.py:5-73 (proximal optimiser; the reference
// differentiates with autograd, here the gradient is written out).
//
// Nothing of size (L, L, 14, 14) is ever stored: one wave64 owns residue i, culls partner residues by
// bounding spheres (exact: the hinge of a culled pair is identically zero), and walks the surviving
// 14x14 atom pairs out of L1/L2-resident coordinates.  Each wave also accumulates dLoss/dxyz of its
// own atoms (a gather formulation, so no atomics and a run-to-run reproducible sum) and folds it into
// dLoss/dchi with d p/d chi_k = axis_k x (p - origin_k) for atoms downstream of chi_k.
#include "pp_internal.h"
#include <math.h>

struct M3 { float m[9]; };
struct Rig { M3 R; float t[3]; };

__device__ __forceinline__ M3 mul33(const M3 &a, const M3 &b) {
    M3 o;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            o.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return o;
}
__device__ __forceinline__ void rot3(const M3 &a, const float *v, float *o) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = a.m[3 * i] * v[0] + a.m[3 * i + 1] * v[1] + a.m[3 * i + 2] * v[2];
}
__device__ __forceinline__ Rig compose(const Rig &a, const Rig &b) {
    Rig o;
    o.R = mul33(a.R, b.R);
    float rt[3];
    rot3(a.R, b.t, rt);
#pragma unroll
    for (int i = 0; i < 3; i++) o.t[i] = rt[i] + a.t[i];
    return o;
}

// default frame g of residue type S composed with the torsion rotation about x:  D_g * Rx(sin, cos)
// rows 0..2 of default frame g (4x4, row-major): three 16-byte loads, issued where this is called
struct DF {
    float4 r[3];
};
__device__ __forceinline__ DF load_df(const float *__restrict__ df, int g) {
    const float4 *p = reinterpret_cast<const float4 *>(df + g * 16);
    return DF{{p[0], p[1], p[2]}};
}
__device__ __forceinline__ Rig torsion_frame(const DF &f, float sn, float cs) {
    Rig D, o;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        D.R.m[3 * i] = f.r[i].x; D.R.m[3 * i + 1] = f.r[i].y; D.R.m[3 * i + 2] = f.r[i].z;
        D.t[i] = f.r[i].w;
    }
    const M3 Rx = {{1.f, 0.f, 0.f, 0.f, cs, -sn, 0.f, sn, cs}};
    o.R = mul33(D.R, Rx);
#pragma unroll
    for (int i = 0; i < 3; i++) o.t[i] = D.t[i];
    return o;
}
__device__ __forceinline__ float sel8(const float (&v)[8], int g) {
    float r = v[0];
#pragma unroll
    for (int k = 1; k < 8; k++) r = g == k ? v[k] : r;
    return r;
}

// 16 lanes per residue: lane a < 14 builds atom a (its own rigid-group frame: the chain chi1 .. chi_g is composed by
// every lane up to its group), lane 14 walks the whole chain and emits the four chi axes, and the per-residue
// reductions (bounding spheres, side-chain atom count) are 16-lane butterflies.  (One thread per residue took 26 us
// at T1124 -- a 3000-instruction dependent chain on 12 waves; this layout takes ~5.)
// One Adam step of the proximal optimiser (UPD argument of k_clash<CAND, true>): since round 5 it runs in the TAIL of the clash
// kernel, on the residue whose gradient that workgroup has just reduced, and goes straight on to the reconstruction at the new
// angles (one launch per Adam step instead of two).  loss_t is parked per residue in `loss_part` and summed in the fixed order of
// k_prox_losses, so the loss curve does not depend on arrival order.
struct ProxUpd {
    int t, nblocks;
    float lamda, step_size, bc2s, inv_n;
    const float *chi0, *z;
    const uint8_t *mask;
    float *x, *m, *v, *xeff, *traj, *last, *loss_part;
};

// What the reconstruction of residue n needs on lane a (0..15) and does not depend on the angles: requested in one batch, so that
// every memory round trip of the chain starts before the first dependent instruction.
struct A14In {
    int S, g, g0;
    float ca[3];
    Rig G;
    float lp[3], am, ex_t, br_t;
    DF f0, f5, f6, f7;
    float xa_t[3];
    float bbd_t;
};
__device__ __forceinline__ A14In a14_load(int n, int a, int S, const float *__restrict__ X,
                                          const float *__restrict__ BB_D, const float *__restrict__ default_frames,
                                          const int32_t *__restrict__ a2g, const float *__restrict__ amask14,
                                          const float *__restrict__ lit, const float *__restrict__ atom_exists,
                                          const float *__restrict__ between_radius) {
    A14In I;
    I.S = S;
    const float *x = X + (size_t)n * 42;
    // backbone frame (same construction as k_frames)
    float av[3], bv[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { I.ca[k] = x[3 + k]; av[k] = x[6 + k] - I.ca[k]; bv[k] = x[k] - I.ca[k]; }
    float na = sqrtf(av[0] * av[0] + av[1] * av[1] + av[2] * av[2] + 1e-8f);
#pragma unroll
    for (int k = 0; k < 3; k++) av[k] /= na;
    float dot = av[0] * bv[0] + av[1] * bv[1] + av[2] * bv[2];
#pragma unroll
    for (int k = 0; k < 3; k++) bv[k] -= av[k] * dot;
    float nb = sqrtf(bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2] + 1e-8f);
#pragma unroll
    for (int k = 0; k < 3; k++) bv[k] /= nb;
    float cv[3] = {av[1] * bv[2] - av[2] * bv[1], av[2] * bv[0] - av[0] * bv[2], av[0] * bv[1] - av[1] * bv[0]};
#pragma unroll
    for (int r = 0; r < 3; r++) { I.G.R.m[3 * r] = av[r]; I.G.R.m[3 * r + 1] = bv[r]; I.G.R.m[3 * r + 2] = cv[r]; I.G.t[r] = I.ca[r]; }

    const float *df = default_frames + (size_t)S * 8 * 16;
    I.g = a < 14 ? a2g[S * 14 + a] : (a == 14 ? 7 : 0);     // lane 14 walks the full chi chain
    // per-atom table entries: fetched here, unconditionally (lanes 14, 15 mirror atom 13), not inside the branches that use them
    const int ac = a < 14 ? a : 13;
    const float *lp_ = lit + ((size_t)S * 14 + ac) * 3;
    I.lp[0] = lp_[0]; I.lp[1] = lp_[1]; I.lp[2] = lp_[2];
    I.am = amask14[S * 14 + ac];
    I.ex_t = atom_exists ? atom_exists[(size_t)n * 14 + ac] : 1.f;
    I.br_t = between_radius[S * 14 + ac];
    // the default frames of the lane's own chain: group min(g, 4) and the chi2..chi4 groups 5..7 (fetched where they are used they
    // were four dependent waits)
    I.g0 = I.g < 4 ? I.g : 4;
    I.f0 = load_df(df, I.g0); I.f5 = load_df(df, 5); I.f6 = load_df(df, 6); I.f7 = load_df(df, 7);
    I.xa_t[0] = x[3 * (a < 4 ? a : 3)]; I.xa_t[1] = x[3 * (a < 4 ? a : 3) + 1]; I.xa_t[2] = x[3 * (a < 4 ? a : 3) + 2];
    I.bbd_t = BB_D[(size_t)n * 3 + (a < 3 ? a : 2)];
    return I;
}
// The chain itself: `ang` = the angle lane a < 7 of the residue's 16 lanes evaluates (phi-like 0..2 from BB_D, chi 3..6).  The 16
// lanes of a residue must be 16 consecutive lanes of one wave (the 16-wide shuffles).
__device__ __forceinline__ void a14_finish(const A14In &I, int n, int a, bool live, float ang, const int64_t *__restrict__ rindex,
                                           float *__restrict__ xyz, float *__restrict__ axes, float *__restrict__ brad,
                                           float4 *__restrict__ rec) {
    const int g = I.g, g0 = I.g0, S = I.S;
    const Rig &G = I.G;
    // the 7 angles as normalised (sin, cos): lane k < 7 evaluates angle k; group g uses angle g-1
    float my_s = 0.f, my_c = 1.f;
    if (a < 7) {
        const float s0 = sinf(ang), c0 = cosf(ang);
        const float den = sqrtf(fmaxf(s0 * s0 + c0 * c0, 1e-12f));
        my_s = s0 / den; my_c = c0 / den;
    }
    float sn[8], cs[8];
    sn[0] = 0.f; cs[0] = 1.f;
#pragma unroll
    for (int gq = 1; gq < 8; gq++) { sn[gq] = __shfl(my_s, gq - 1, 16); cs[gq] = __shfl(my_c, gq - 1, 16); }

    Rig chain = torsion_frame(I.f0, sel8(sn, g0), sel8(cs, g0));
    const bool axis_lane = a == 14 && axes != nullptr && live;
    float ax[4][6];                       // lane 14: the four chi axes in global coordinates (stored below, one branch)
    {
        const Rig Fg = compose(G, chain);
        ax[0][0] = Fg.R.m[0]; ax[0][1] = Fg.R.m[3]; ax[0][2] = Fg.R.m[6]; ax[0][3] = Fg.t[0]; ax[0][4] = Fg.t[1]; ax[0][5] = Fg.t[2];
    }
#pragma unroll
    for (int gg = 5; gg < 8; gg++) {
        const Rig nx = compose(chain, torsion_frame(gg == 5 ? I.f5 : gg == 6 ? I.f6 : I.f7, sn[gg], cs[gg]));
        const bool take = gg <= g;        // selects, not a branch: the chain stays one basic block
#pragma unroll
        for (int q = 0; q < 9; q++) chain.R.m[q] = take ? nx.R.m[q] : chain.R.m[q];
#pragma unroll
        for (int q = 0; q < 3; q++) chain.t[q] = take ? nx.t[q] : chain.t[q];
        const Rig Fg = compose(G, chain);
        ax[gg - 4][0] = Fg.R.m[0]; ax[gg - 4][1] = Fg.R.m[3]; ax[gg - 4][2] = Fg.R.m[6];
        ax[gg - 4][3] = Fg.t[0]; ax[gg - 4][4] = Fg.t[1]; ax[gg - 4][5] = Fg.t[2];
    }
    if (axis_lane) {
#pragma unroll
        for (int c4 = 0; c4 < 4; c4++)
#pragma unroll
            for (int q = 0; q < 6; q++) axes[((size_t)n * 4 + c4) * 6 + q] = ax[c4][q];
    }
    const Rig F = compose(G, chain);
    float p[3] = {0.f, 0.f, 0.f}, ex = 0.f, reff = 0.f;
    if (a < 14) {
        if (a < 4) {
#pragma unroll
            for (int k = 0; k < 3; k++) p[k] = I.xa_t[k];
        } else {
            float rp[3];
            rot3(F.R, I.lp, rp);
#pragma unroll
            for (int k = 0; k < 3; k++) p[k] = (rp[k] + F.t[k]) * I.am;
        }
        ex = I.ex_t;
        reff = ex * I.br_t;
        if (live) {
#pragma unroll
            for (int k = 0; k < 3; k++) xyz[((size_t)n * 14 + a) * 3 + k] = p[k];
            if (rec) rec[(size_t)n * 16 + a] = make_float4(p[0], p[1], p[2], reff);
        }
    }
    // per-residue reductions over the 16 lanes
    const float *ca = I.ca;
    const bool has = a < 14 && ex != 0.f;
    float dca = 0.f;
    if (has) { float dx = p[0] - ca[0], dy = p[1] - ca[1], dz = p[2] - ca[2]; dca = dx * dx + dy * dy + dz * dz; }
    float cnt = has ? 1.f : 0.f, sx = has ? p[0] : 0.f, sy = has ? p[1] : 0.f, sz = has ? p[2] : 0.f;
    float nsc = (a >= 4 && a < 14) ? ex : 0.f;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        dca = fmaxf(dca, __shfl_xor(dca, o, 16));
        cnt += __shfl_xor(cnt, o, 16); sx += __shfl_xor(sx, o, 16); sy += __shfl_xor(sy, o, 16); sz += __shfl_xor(sz, o, 16);
        nsc += __shfl_xor(nsc, o, 16);
    }
    const float inv = 1.f / fmaxf(cnt, 1.f);
    const float cen[3] = {sx * inv, sy * inv, sz * inv};
    float r2 = 0.f;
    if (has) { float dx = p[0] - cen[0], dy = p[1] - cen[1], dz = p[2] - cen[2]; r2 = dx * dx + dy * dy + dz * dz; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o, 16));
    if (live && a == 0) {
        if (brad) brad[n] = sqrtf(dca) * 1.0001f + 1e-3f;
        if (rec) {
            // bounding sphere about the centroid of the atoms present (tighter than about CA: ~2.5x fewer candidates)
            rec[(size_t)n * 16 + 14] = make_float4(cen[0], cen[1], cen[2], sqrtf(r2) * 1.0001f + 1e-3f);
            rec[(size_t)n * 16 + 15] = make_float4(nsc, __int_as_float((int)rindex[n]), __int_as_float(S), 0.f);
        }
    }
}

__global__ void __launch_bounds__(256)
k_atom14(int N, const float *__restrict__ X, const int64_t *__restrict__ rtype,
         const float *__restrict__ BB_D, const float *__restrict__ chi,
         const float *__restrict__ default_frames, const int32_t *__restrict__ a2g,
         const float *__restrict__ amask14, const float *__restrict__ lit,
         const float *__restrict__ atom_exists, const float *__restrict__ between_radius,
         const int64_t *__restrict__ rindex,
         float *__restrict__ xyz, float *__restrict__ axes, float *__restrict__ brad,
         float4 *__restrict__ rec) {
    const int a = threadIdx.x & 15;
    const int nraw = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = nraw < N;
    const int n = live ? nraw : N - 1;           // out-of-range groups mirror the last residue and store nothing
    const A14In I = a14_load(n, a, (int)rtype[n], X, BB_D, default_frames, a2g, amask14, lit, atom_exists, between_radius);
    const float chi_t = chi[(size_t)n * 4 + (a >= 3 && a < 7 ? a - 3 : 0)];
    a14_finish(I, n, a, live, a < 3 ? I.bbd_t : chi_t, rindex, xyz, axes, brad, rec);
}

// ---------------------------------------------------------------------------------------------
// clash: one wave per residue.  lane = 16 * slot + a,  a = own atom (0..13), slot = 0..3 partner stripe
// ---------------------------------------------------------------------------------------------
#ifndef CL_WAVES
#define CL_WAVES 4
#endif
#define CL_MAXC (8192 / CL_WAVES)     // candidate list capacity per wave (entries beyond are handled by re-scanning)

// CAND (the Adam loop of pp_proximal): the partner residues come from the static candidate lists k_clash_cand built once for the
// whole loop instead of a scan over all L partners of the complex at every step -- the lists hold every residue pair whose hinge can
// be non-zero for ANY chi (the backbone does not move), in the order and under the wave assignment of the scan, and the exact
// per-step sphere test still runs on them: the atom pairs that contribute are the same, every contribution is the same number.
//
// FUSE (the Adam loop of pp_proximal, round 5): wave 0 goes on, for ITS residue, to Adam step U.t with the gradient it has just reduced
// (the operands of optimize.py:47-71 as the separate k_atom14<true> launch of rounds 2-4 read them: per_res and dchi now come from
// registers) and to the reconstruction at the new angles, which it writes into the OTHER record / axes buffer (F.rec_out, F.axes_out:
// the other workgroups of this launch still read this step's) -- one launch per Adam step instead of two.  The four partner stripes of
// wave 0 run the tail redundantly (same addresses, same values; stripe 0 stores): no divergence inside the 16-lane shuffles.
struct ClashFuse {
    const float *X, *BB_D, *default_frames, *amask14, *lit, *atom_exists, *between_radius;
    const int64_t *rindex;
    float *xyz, *axes_out, *brad;
    float4 *rec_out;
    ProxUpd U;
};
// (The tail's reconstruction chain wants ~130 registers; with the 33 KB candidate lists the compiler holds the kernel to the 128 of four
// waves per SIMD and spills twenty dwords in the tail.  Buying them with amdgpu_waves_per_eu(1, 3) -- 129 registers, no scratch -- made
// the launch 2.2x SLOWER, 36 instead of 16 us at T1124: profiles/r05_prox_fused_step.txt.  The spill stays.)
template <bool CAND, bool FUSE>
__global__ void __launch_bounds__(64 * CL_WAVES)
k_clash(int N, const int2 *__restrict__ seg, const float *__restrict__ xyz, const float4 *__restrict__ rec, const float *__restrict__ exists,
        const float *__restrict__ lower, const float *__restrict__ upper, const int32_t *__restrict__ a2g,
        const float *__restrict__ axes, float tol, float inv_ntot,
        float *__restrict__ per_res, float *__restrict__ dchi, const int32_t *__restrict__ cand, const int32_t *__restrict__ cand_cnt,
        ClashFuse F) {
    __shared__ int s_list[CL_WAVES][CL_MAXC];
    __shared__ float s_red[CL_WAVES][16][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x;              // one workgroup per residue; its waves take interleaved 64-partner windows
    if (i >= N) return;
    const int row0 = seg[i].x, L = seg[i].y;      // partner residues: the rows of this residue's own complex
    const int a = lane & 15, slot = lane >> 4;
    const bool own = a < 14;
    // rec[n] (written by k_atom14): 14 x (x, y, z, exists * radius) | (CA, bounding radius) | (n side-chain atoms,
    // residue_index, residue type): every partner needs 15 coalesced 16-byte reads instead of ~85 scattered ones
    const float4 me = rec[(size_t)i * 16 + 15];
    const int S = __float_as_int(me.z);
    const int ri = __float_as_int(me.y);
    float pa[3] = {0.f, 0.f, 0.f}, ea = 0.f, ra = 0.f;
    if (own) {
        const float4 q = rec[(size_t)i * 16 + a];
        pa[0] = q.x; pa[1] = q.y; pa[2] = q.z;
        ra = q.w;
        ea = q.w != 0.f ? 1.f : 0.f;
    }
    const float nsc = me.x;                          // number of side-chain atoms -> this residue's weight in the mean
    const float wi = inv_ntot / (nsc + 1e-10f);
    const float4 cme = rec[(size_t)i * 16 + 14];          // bounding sphere (centroid, radius)
    const float cai[3] = {cme.x, cme.y, cme.z};
    const float radi = cme.w;
    const float reach = 3.6f - tol;                 // largest r_a + r_b - tol (S-S)

    float loss_a = 0.f, ga[3] = {0.f, 0.f, 0.f};
    int *list = s_list[wave];
    int n_static = -1;
    if constexpr (CAND) n_static = cand_cnt[(size_t)i * CL_WAVES + wave];
    const int32_t *my_cand = CAND ? cand + ((size_t)i * CL_WAVES + wave) * PP_CL_CAP : nullptr;
    // partner residues of the same complex: from the static candidates (one pass), or in windows that fit the candidate list
    const bool use_static = CAND && n_static >= 0;
    bool more = true;
    for (int base = 64 * wave; use_static ? more : base < L; ) {
        int cnt = 0;
        int jscan = base;
        if (use_static) {
            // the static candidates of this wave (ascending, at most PP_CL_CAP <= CL_MAXC): the exact sphere test on each, compacted in order
            for (int c0 = 0; c0 < n_static; c0 += 64) {
                const int ci = c0 + lane;
                bool keep = false;
                int jg = 0;
                if (ci < n_static) {
                    jg = my_cand[ci];
                    const float4 cj = rec[(size_t)jg * 16 + 14];
                    float dx = cj.x - cai[0], dy = cj.y - cai[1], dz = cj.z - cai[2];
                    float lim = radi + cj.w + reach;
                    keep = (lim > 0.f) && (dx * dx + dy * dy + dz * dz < lim * lim);
                }
                unsigned long long bal = __ballot(keep);
                if (keep) list[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = jg;
                cnt += __popcll(bal);
            }
            more = false;
        } else {
            for (; jscan < L && cnt + 64 <= CL_MAXC; jscan += 64 * CL_WAVES) {
                int jl = jscan + lane;
                bool keep = false;
                if (jl < L) {
                    int jg = row0 + jl;
                    if (jg != i) {
                        const float4 cj = rec[(size_t)jg * 16 + 14];
                        const float4 mj = rec[(size_t)jg * 16 + 15];
                        float dx = cj.x - cai[0], dy = cj.y - cai[1], dz = cj.z - cai[2];
                        float lim = radi + cj.w + reach;
                        keep = (lim > 0.f) && (dx * dx + dy * dy + dz * dz < lim * lim) && (__float_as_int(mj.y) != ri);
                    }
                }
                unsigned long long bal = __ballot(keep);
                if (keep) list[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = row0 + jl;
                cnt += __popcll(bal);
            }
        }
        base = jscan;
        __builtin_amdgcn_wave_barrier();
        for (int c = slot; c < cnt; c += 4) {
            const int jg = list[c];
            // all of the partner's records first, unconditionally: inside the branches below the compiler may not hoist them,
            // and fourteen dependent round trips per candidate were 13 of this kernel's 20 us at T1124 (fetching the next
            // candidate's records one iteration ahead on top of this gains nothing)
            float4 pbr[14];
#pragma unroll
            for (int bb = 0; bb < 14; bb++) pbr[bb] = rec[(size_t)jg * 16 + bb];
            const float4 mj = rec[(size_t)jg * 16 + 15];
            const int rj = __float_as_int(mj.y);
            const bool i_low = ri < rj;
            const bool adjacent = i_low ? (ri + 1 == rj) : (rj + 1 == ri);
            const float wj = inv_ntot / (mj.x + 1e-10f);
            if (own && ea != 0.f) {
#pragma unroll
                for (int bb = 0; bb < 14; bb++) {
                    const float4 pb = pbr[bb];
                    bool ok = pb.w != 0.f && !(a < 4 && bb < 4) && !(a == 5 && bb == 5);
                    if (adjacent) {
                        // peptide bond C(lower) - N(higher)
                        if (i_low ? (a == 2 && bb == 0) : (a == 0 && bb == 2)) ok = false;
                    }
                    if (ok) {
                        float dx = pa[0] - pb.x, dy = pa[1] - pb.y, dz = pa[2] - pb.z;
                        // squared test first: the IEEE sqrt and the division below are ~50 instructions, and only a few
                        // per cent of the surviving atom pairs overlap (most trips skip the branch for the whole wave)
                        const float d2 = 1e-10f + dx * dx + dy * dy + dz * dz;
                        const float thr = (ra + pb.w) - tol;
                        if (!(thr > 0.f && d2 < thr * thr)) continue;
                        float d = sqrtf(d2);
                        float err = thr - d;
                        if (err > 0.f) {
                            loss_a += err;
                            float cw = (a >= 4 ? wi : 0.f) + (bb >= 4 ? wj : 0.f);
                            float sc = -cw / d;
                            ga[0] = fmaf(sc, dx, ga[0]); ga[1] = fmaf(sc, dy, ga[1]); ga[2] = fmaf(sc, dz, ga[2]);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // fold the 4 partner stripes, then the 4 waves (fixed order: reproducible)
    for (int o = 16; o <= 32; o <<= 1) {
        loss_a += __shfl_xor(loss_a, o);
        ga[0] += __shfl_xor(ga[0], o); ga[1] += __shfl_xor(ga[1], o); ga[2] += __shfl_xor(ga[2], o);
    }
    if (lane < 16) {
        s_red[wave][lane][0] = loss_a; s_red[wave][lane][1] = ga[0]; s_red[wave][lane][2] = ga[1]; s_red[wave][lane][3] = ga[2];
    }
    __syncthreads();
    if (wave != 0) return;
    if (slot == 0) {
#pragma unroll
        for (int w = 1; w < CL_WAVES; w++) {
            loss_a += s_red[w][a][0]; ga[0] += s_red[w][a][1]; ga[1] += s_red[w][a][2]; ga[2] += s_red[w][a][3];
        }
    } else {
        loss_a = 0.f; ga[0] = ga[1] = ga[2] = 0.f;
    }
    // within-residue bounds: stripes of partner atoms b = slot, slot+4, ...
    if (own && ea != 0.f) {
        for (int bb = slot; bb < 14; bb += 4) {
            if (bb == a || (a < 4 && bb < 4)) continue;
            const float4 pb = rec[(size_t)i * 16 + bb];
            if (pb.w == 0.f) continue;
            float dx = pa[0] - pb.x, dy = pa[1] - pb.y, dz = pa[2] - pb.z;
            float d = sqrtf(1e-10f + dx * dx + dy * dy + dz * dz);
            float lo = lower[(S * 14 + a) * 14 + bb], up = upper[(S * 14 + a) * 14 + bb];
            float e_lo = lo - d, e_up = d - up;
            float l = fmaxf(e_lo, 0.f) + fmaxf(e_up, 0.f);
            loss_a += 2.f * l;                                  // row sum + column sum of a symmetric table
            float dl = (e_up > 0.f ? 1.f : 0.f) - (e_lo > 0.f ? 1.f : 0.f);
            float cw = 2.f * ((a >= 4 ? wi : 0.f) + (bb >= 4 ? wi : 0.f));
            float sc = cw * dl / d;
            ga[0] = fmaf(sc, dx, ga[0]); ga[1] = fmaf(sc, dy, ga[1]); ga[2] = fmaf(sc, dz, ga[2]);
        }
    }
    // fold the 4 partner stripes
    for (int o = 16; o <= 32; o <<= 1) {
        loss_a += __shfl_xor(loss_a, o);
        ga[0] += __shfl_xor(ga[0], o); ga[1] += __shfl_xor(ga[1], o); ga[2] += __shfl_xor(ga[2], o);
    }
    float lres = (own && a >= 4) ? loss_a : 0.f;
    for (int o = 8; o > 0; o >>= 1) lres += __shfl_xor(lres, o);
    const float pres = lres / (nsc + 1e-10f);
    if (lane == 0) per_res[i] = pres;
    float dk[4] = {0.f, 0.f, 0.f, 0.f};
    if (dchi || FUSE) {
        if (own && a >= 5) {
            const int g = a2g[S * 14 + a];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (g >= 4 + k) {
                    const float *ax = axes + ((size_t)i * 4 + k) * 6;
                    float rx = pa[0] - ax[3], ry = pa[1] - ax[4], rz = pa[2] - ax[5];
                    float cx = ax[1] * rz - ax[2] * ry, cy = ax[2] * rx - ax[0] * rz, cz = ax[0] * ry - ax[1] * rx;
                    dk[k] = ga[0] * cx + ga[1] * cy + ga[2] * cz;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            for (int o = 8; o > 0; o >>= 1) dk[k] += __shfl_xor(dk[k], o);
        }
        if (dchi && lane < 4) dchi[(size_t)i * 4 + lane] = lane == 0 ? dk[0] : (lane == 1 ? dk[1] : (lane == 2 ? dk[2] : dk[3]));
    }
    if constexpr (FUSE) {
        const ProxUpd &U = F.U;
        const bool st = slot == 0;                  // the stripe that stores
        // everything the tail reads, in one batch: the reconstruction's inputs and the Adam operands of chi_k, k = a < 4
        const A14In I = a14_load(i, a, S, F.X, F.BB_D, F.default_frames, a2g, F.amask14, F.lit, F.atom_exists, F.between_radius);
        const int k = a < 4 ? a : 0;
        const size_t e = (size_t)i * 4 + k;
        const float xe = U.xeff[e], ze = U.z[e], xo = U.x[e], mo = U.m[e], vo = U.v[e], c0v = U.chi0[e];
        const bool mk = U.mask[i] != 0;
        // loss_t = mean_n [sum_k (xeff - z)^2 + lamda per_res] at the incoming iterate; then torch.optim.Adam defaults (lr 1e-2,
        // betas (0.9, 0.999), eps 1e-8, bias-corrected; step_size = lr / (1 - beta1^t) and bc2s = sqrt(1 - beta2^t) come from the
        // host in double); outputs as optimize.py:66-71
        const float dch = k == 0 ? dk[0] : (k == 1 ? dk[1] : (k == 2 ? dk[2] : dk[3]));
        float q = 0.f, outv = 0.f;
        if (a < 4) {
            const float d = xe - ze;
            q = fabsf(d) * fabsf(d);
            const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
            float g = 0.f;
            if (mk) g = 2.f * (xo - ze) * U.inv_n + U.lamda * dch;
            const float mm = mo + (g - mo) * (1.f - b1);             // exp_avg.lerp_(grad, 1 - beta1)
            const float vv = vo * b2 + (1.f - b2) * (g * g);
            const float denom = sqrtf(vv) / U.bc2s + eps;
            const float xn = xo - U.step_size * (mm / denom);
            outv = mk ? xn : c0v;
            if (st) {
                U.m[e] = mm; U.v[e] = vv; U.x[e] = xn;
                U.xeff[e] = outv;
                if (U.traj) U.traj[(size_t)U.t * N * 4 + e] = outv;
                if (U.last) U.last[e] = outv;
            }
        }
        q += __shfl_xor(q, 1, 16);
        q += __shfl_xor(q, 2, 16);
        if (lane == 0) U.loss_part[(size_t)U.t * N + i] = q + U.lamda * pres;
        // the reconstruction at the new angles: lane a in 3..6 evaluates chi_(a-3), which lane a - 3 has just stepped
        const float chi_new = __shfl(outv, (a - 3) & 15, 16);
        a14_finish(I, i, a, st, a < 3 ? I.bbd_t : chi_new, F.rindex, F.xyz, F.axes_out, F.brad, F.rec_out);
    }
}

// ---------------------------------------------------------------------------------------------
// static clash-partner candidates of the proximal loop: once per pp_proximal
// ---------------------------------------------------------------------------------------------
// The Adam loop moves side chains, never the backbone.  An atom of residue i can only ever overlap an atom of residue j if
//   |CA_i - CA_j| < e_i + e_j + (3.6 - tol)      e = how far an atom of the residue can be from its CA for ANY chi:
// the side-chain bound of the residue type (plan->side_extent) or the actual distance of its N / C / O, whichever is larger.  Every
// other pair has a zero hinge at every step, so k_clash<true> need not look at it: instead of scanning the L partners of the
// complex at every step (O(L^2) sphere tests per launch: 98 % of them culled at T1124, 99 % at S1500) a wave reads its few dozen
// candidates.  Same workgroup shape and wave / window assignment as the scan of k_clash: wave w of residue i's workgroup gets the
// partners of windows 64 w + 256 m, ascending; more than PP_CL_CAP of them and the wave keeps the full scan (count -1).
__global__ void __launch_bounds__(64 * CL_WAVES)
k_clash_cand(int N, const int2 *__restrict__ seg, const float *__restrict__ X, const float *__restrict__ amask, const int64_t *__restrict__ rtype,
             const int64_t *__restrict__ rindex, const float *__restrict__ side_extent, float tol,
             int32_t *__restrict__ cand, int32_t *__restrict__ cand_cnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x;
    if (i >= N) return;
    const int row0 = seg[i].x, L = seg[i].y;
    auto extent = [&](int n, float (&ca)[3]) {
        const float *x = X + (size_t)n * 42;
        ca[0] = x[3]; ca[1] = x[4]; ca[2] = x[5];
        float e = side_extent[(int)rtype[n]];
        const float *m = amask + (size_t)n * 14;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            if (a == 1 || m[a] == 0.f) continue;
            const float dx = x[3 * a] - ca[0], dy = x[3 * a + 1] - ca[1], dz = x[3 * a + 2] - ca[2];
            e = fmaxf(e, sqrtf(dx * dx + dy * dy + dz * dz) * 1.0001f + 1e-3f);
        }
        return e;
    };
    float cai[3];
    const float ei = extent(i, cai);
    const int ri = (int)rindex[i];
    const float reach = 3.6f - tol;
    int32_t *out = cand + ((size_t)i * CL_WAVES + wave) * PP_CL_CAP;
    int cnt = 0;
    for (int jscan = 64 * wave; jscan < L; jscan += 64 * CL_WAVES) {
        const int jl = jscan + lane;
        bool keep = false;
        if (jl < L) {
            const int jg = row0 + jl;
            if (jg != i && (int)rindex[jg] != ri) {
                float caj[3];
                const float ej = extent(jg, caj);
                const float dx = caj[0] - cai[0], dy = caj[1] - cai[1], dz = caj[2] - cai[2];
                const float lim = ei + ej + reach;
                keep = lim > 0.f && dx * dx + dy * dy + dz * dz < lim * lim;
            }
        }
        const unsigned long long bal = __ballot(keep);
        const int at = cnt + __popcll(bal & ((1ull << lane) - 1ull));
        if (keep && at < PP_CL_CAP) out[at] = row0 + jl;
        cnt += __popcll(bal);
    }
    if (lane == 0) cand_cnt[(size_t)i * CL_WAVES + wave] = cnt <= PP_CL_CAP ? cnt : -1;
}

// ---------------------------------------------------------------------------------------------
// proximal optimiser pieces (B = 1)
// ---------------------------------------------------------------------------------------------
// mean of per_res -> scal[0]; mask[n] = per_res[n] > mean; z = chi*mask; x = z; m = v = 0; xeff = chi
__global__ void __launch_bounds__(1024)
k_prox_init(int N, const float *__restrict__ per_res, const float *__restrict__ chi, uint8_t *__restrict__ mask,
            float *__restrict__ z, float *__restrict__ x, float *__restrict__ m, float *__restrict__ v,
            float *__restrict__ xeff) {
    __shared__ float s_part[16];
    __shared__ float s_mean;
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 1024) s += per_res[n];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; w++) t += s_part[w];
        s_mean = t / (float)N;
    }
    __syncthreads();
    const float mean = s_mean;
    for (int n = threadIdx.x; n < N; n += 1024) {
        const bool mk = per_res[n] > mean;
        mask[n] = mk ? 1 : 0;
        for (int k = 0; k < 4; k++) {
            const float c = chi[(size_t)n * 4 + k];
            const float zz = mk ? c : 0.f;
            z[(size_t)n * 4 + k] = zz;
            x[(size_t)n * 4 + k] = zz;
            m[(size_t)n * 4 + k] = 0.f;
            v[(size_t)n * 4 + k] = 0.f;
            xeff[(size_t)n * 4 + k] = c;
        }
    }
}

// losses[t0 + t] = (1 / N) sum of the per-residue terms the fused clash kernel left, in a fixed order: groups of 16 residues first,
// then the groups in order (the order of rounds 2-4, when a 16-residue block of k_atom14<true> summed its own terms).  One workgroup per
// step: a lane adds up one group (16 sequential adds), the group sums meet in LDS and lane 0 adds them in order -- the chain of N dependent
// loads a single lane per step would walk is 185 us at T1124.
__global__ void __launch_bounds__(256)
k_prox_losses(int N, float inv_n, const float *__restrict__ part, float *__restrict__ losses) {
    __shared__ float s_tt[256];
    const int t = blockIdx.x;
    const int ngroups = (N + 15) / 16;
    float s = 0.f;
    for (int g0 = 0; g0 < ngroups; g0 += 256) {          // 256 groups (4096 residues) at a time, in order
        const int g = g0 + threadIdx.x;
        float tt = 0.f;
        if (g < ngroups)
            for (int i = 16 * g; i < 16 * g + 16 && i < N; i++) tt += part[(size_t)t * N + i];
        s_tt[threadIdx.x] = tt;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int n = ngroups - g0 < 256 ? ngroups - g0 : 256;
            for (int q = 0; q < n; q++) s += s_tt[q];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) losses[t] = s * inv_n;
}

pp_status pp_launch_atom14(pp_ctx *c, const float *chi, float *xyz, hipStream_t s) {
    const pp_plan *p = c->plan;
    // the packed records feed k_clash; they need the residue numbering, which geometry-only batches may not carry
    float4 *rec = c->b.residue_index ? reinterpret_cast<float4 *>(c->rec) : nullptr;
    hipLaunchKernelGGL(k_atom14, dim3((c->N + 15) / 16), dim3(256), 0, s, c->N, c->b.X, c->b.residue_type, c->b.BB_D, chi,
                       p->default_frames, p->atom14_to_group, p->atom14_mask, p->lit_positions, c->b.atom_mask,
                       p->between_radius, c->b.residue_index, xyz, c->axes, c->brad, rec);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_clash(pp_ctx *c, const float *xyz, float *per_res, float *dchi, hipStream_t s, bool use_candidates) {
    const pp_plan *p = c->plan;
    if (use_candidates && c->cand)
        PP_LAUNCH(c, (k_clash<true, false>), dim3(c->N), dim3(64 * CL_WAVES), 0, s, c->N, c->seg, xyz,
                  reinterpret_cast<const float4 *>(c->rec), c->b.atom_mask,
                  p->bounds_lower, p->bounds_upper, p->atom14_to_group, c->axes, p->clash_tol,
                  1.0f / (float)c->N, per_res, dchi, c->cand, c->cand_cnt, ClashFuse{});
    else
        PP_LAUNCH(c, (k_clash<false, false>), dim3(c->N), dim3(64 * CL_WAVES), 0, s, c->N, c->seg, xyz,
                  reinterpret_cast<const float4 *>(c->rec), c->b.atom_mask,
                  p->bounds_lower, p->bounds_upper, p->atom14_to_group, c->axes, p->clash_tol,
                  1.0f / (float)c->N, per_res, dchi, nullptr, nullptr, ClashFuse{});
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_proximal(pp_ctx *c, const float *chi, float lamda, int nsteps, float *traj, float *chi_last,
                             float *losses, hipStream_t s) {
    pp_status st;
    const pp_plan *p = c->plan;
    // static partner candidates of the whole loop (the backbone does not move): k_clash<true> reads them instead of scanning
    static const char *scan_env = PP_GETENV("PP_CLASH_SCAN");          // diagnostic builds: 1 = the per-step scan of all partners (A/B)
    const bool cands = c->cand != nullptr && !(scan_env && atoi(scan_env));
    if (cands)
        hipLaunchKernelGGL(k_clash_cand, dim3(c->N), dim3(64 * CL_WAVES), 0, s, c->N, c->seg, c->b.X, c->b.atom_mask, c->b.residue_type,
                           c->b.residue_index, p->side_extent, p->clash_tol, c->cand, c->cand_cnt);
    // clash mask at the incoming angles (optimize.py:5-18)
    if ((st = pp_launch_atom14(c, chi, c->xyz, s)) != PP_OK) return st;
    if ((st = pp_launch_clash(c, c->xyz, c->per_res, nullptr, s, cands)) != PP_OK) return st;
    hipLaunchKernelGGL(k_prox_init, dim3(1), dim3(1024), 0, s, c->N, c->per_res, chi, c->pmask, c->pz, c->px, c->pm,
                       c->pv, c->pxeff);
    // ONE launch per Adam step: [clash + gradient at the current angles -> step t on the workgroup's own residue -> its
    // reconstruction at the new angles, into the other record / axes buffer].  Loss terms are parked per residue and reduced in a
    // fixed order, PP_PROX_CHUNK steps at a time.
    // (records and axes at the start angles are in c->rec / c->axes already: the atom14 launch above ran at `chi`, which is what
    // k_prox_init has just copied into xeff, and the clash launch between them writes neither)
    float *rec_in = c->rec, *rec_out = c->rec2, *axes_in = c->axes, *axes_out = c->axes2;
    for (int t = 0; t < nsteps; t++) {
        const double bc1 = 1.0 - pow(0.9, (double)(t + 1)), bc2 = 1.0 - pow(0.999, (double)(t + 1));
        ClashFuse F;
        F.X = c->b.X; F.BB_D = c->b.BB_D; F.default_frames = p->default_frames; F.amask14 = p->atom14_mask; F.lit = p->lit_positions;
        F.atom_exists = c->b.atom_mask; F.between_radius = p->between_radius; F.rindex = c->b.residue_index;
        F.xyz = c->xyz; F.axes_out = axes_out; F.brad = c->brad; F.rec_out = reinterpret_cast<float4 *>(rec_out);
        ProxUpd &U = F.U;
        U.nblocks = 0;
        U.lamda = lamda; U.step_size = (float)(1e-2 / bc1); U.bc2s = (float)sqrt(bc2); U.inv_n = 1.0f / (float)c->N;
        U.chi0 = chi; U.z = c->pz; U.mask = c->pmask;
        U.x = c->px; U.m = c->pm; U.v = c->pv; U.xeff = c->pxeff; U.last = chi_last;
        U.loss_part = c->prox_part;
        U.t = t % PP_PROX_CHUNK;
        U.traj = traj ? traj + (size_t)(t - U.t) * c->N * 4 : nullptr;     // U.t indexes within the chunk
        c->prof_armed = c->prof_which == 3;          // pp_profile_kernel(3): the fused clash + Adam step + reconstruction
        if (cands)
            PP_LAUNCH(c, (k_clash<true, true>), dim3(c->N), dim3(64 * CL_WAVES), 0, s, c->N, c->seg, c->xyz,
                      reinterpret_cast<const float4 *>(rec_in), c->b.atom_mask, p->bounds_lower, p->bounds_upper, p->atom14_to_group,
                      axes_in, p->clash_tol, 1.0f / (float)c->N, c->per_res, c->dchi, c->cand, c->cand_cnt, F);
        else
            PP_LAUNCH(c, (k_clash<false, true>), dim3(c->N), dim3(64 * CL_WAVES), 0, s, c->N, c->seg, c->xyz,
                      reinterpret_cast<const float4 *>(rec_in), c->b.atom_mask, p->bounds_lower, p->bounds_upper, p->atom14_to_group,
                      axes_in, p->clash_tol, 1.0f / (float)c->N, c->per_res, c->dchi, nullptr, nullptr, F);
        c->prof_armed = false;
        std::swap(rec_in, rec_out);
        std::swap(axes_in, axes_out);
        const bool chunk_end = U.t == PP_PROX_CHUNK - 1 || t == nsteps - 1;
        if (chunk_end)
            hipLaunchKernelGGL(k_prox_losses, dim3(U.t + 1), dim3(256), 0, s, c->N, U.inv_n, c->prox_part, losses + (t - U.t));
    }
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
