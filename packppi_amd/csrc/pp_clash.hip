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

// one thread per residue
__global__ void k_atom14(int N, const float *__restrict__ X, const int64_t *__restrict__ rtype,
                         const float *__restrict__ BB_D, const float *__restrict__ chi,
                         const float *__restrict__ default_frames, const int32_t *__restrict__ a2g,
                         const float *__restrict__ amask14, const float *__restrict__ lit,
                         const float *__restrict__ atom_exists,
                         float *__restrict__ xyz, float *__restrict__ axes, float *__restrict__ brad) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int S = (int)rtype[n];
    const float *x = X + (size_t)n * 42;
    // backbone frame (same construction as k_frames)
    float a[3], b[3], ca[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { ca[k] = x[3 + k]; a[k] = x[6 + k] - ca[k]; b[k] = x[k] - ca[k]; }
    float na = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + 1e-8f);
#pragma unroll
    for (int k = 0; k < 3; k++) a[k] /= na;
    float dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
#pragma unroll
    for (int k = 0; k < 3; k++) b[k] -= a[k] * dot;
    float nb = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + 1e-8f);
#pragma unroll
    for (int k = 0; k < 3; k++) b[k] /= nb;
    float c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    Rig G;
#pragma unroll
    for (int r = 0; r < 3; r++) { G.R.m[3 * r] = a[r]; G.R.m[3 * r + 1] = b[r]; G.R.m[3 * r + 2] = c[r]; G.t[r] = ca[r]; }

    // the 7 angles as normalised (sin, cos): pre-omega, phi, psi, chi1..4
    float sn[8], cs[8];
    sn[0] = 0.f; cs[0] = 1.f;
#pragma unroll
    for (int k = 0; k < 7; k++) {
        float ang = k < 3 ? BB_D[(size_t)n * 3 + k] : chi[(size_t)n * 4 + (k - 3)];
        float s = sinf(ang), co = cosf(ang);
        float den = sqrtf(fmaxf(s * s + co * co, 1e-12f));
        sn[k + 1] = s / den; cs[k + 1] = co / den;
    }
    const float *df = default_frames + (size_t)S * 8 * 16;
    Rig F[8];
#pragma unroll
    for (int g = 0; g < 8; g++) {
        Rig D;
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int jx = 0; jx < 3; jx++) D.R.m[3 * i + jx] = df[g * 16 + 4 * i + jx];
            D.t[i] = df[g * 16 + 4 * i + 3];
        }
        M3 Rx = {{1.f, 0.f, 0.f, 0.f, cs[g], -sn[g], 0.f, sn[g], cs[g]}};
        F[g].R = mul33(D.R, Rx);
#pragma unroll
        for (int i = 0; i < 3; i++) F[g].t[i] = D.t[i];
    }
    F[5] = compose(F[4], F[5]);
    F[6] = compose(F[5], F[6]);
    F[7] = compose(F[6], F[7]);
#pragma unroll
    for (int g = 0; g < 8; g++) F[g] = compose(G, F[g]);
    // chi-frame rotation axes (x axis of the frame) and origins, for the analytic gradient
    if (axes) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float *o = axes + ((size_t)n * 4 + k) * 6;
            o[0] = F[4 + k].R.m[0]; o[1] = F[4 + k].R.m[3]; o[2] = F[4 + k].R.m[6];
            o[3] = F[4 + k].t[0]; o[4] = F[4 + k].t[1]; o[5] = F[4 + k].t[2];
        }
    }
    float rad2 = 0.f;
    for (int at = 0; at < 14; at++) {
        float p[3];
        if (at < 4) {
#pragma unroll
            for (int k = 0; k < 3; k++) p[k] = x[3 * at + k];
        } else {
            const int g = a2g[S * 14 + at];
            const float *lp = lit + ((size_t)S * 14 + at) * 3;
            const float am = amask14[S * 14 + at];
            Rig Fg = F[0];
#pragma unroll
            for (int gg = 1; gg < 8; gg++) if (g == gg) Fg = F[gg];
            float rp[3];
            rot3(Fg.R, lp, rp);
#pragma unroll
            for (int k = 0; k < 3; k++) p[k] = (rp[k] + Fg.t[k]) * am;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) xyz[((size_t)n * 14 + at) * 3 + k] = p[k];
        if (brad && (!atom_exists || atom_exists[(size_t)n * 14 + at] != 0.f)) {
            float dx = p[0] - ca[0], dy = p[1] - ca[1], dz = p[2] - ca[2];
            rad2 = fmaxf(rad2, dx * dx + dy * dy + dz * dz);
        }
    }
    if (brad) brad[n] = sqrtf(rad2) * 1.0001f + 1e-3f;
}

// ---------------------------------------------------------------------------------------------
// clash: one wave per residue.  lane = 16 * slot + a,  a = own atom (0..13), slot = 0..3 partner stripe
// ---------------------------------------------------------------------------------------------
#define CL_WAVES 4
#define CL_MAXC 2048     // candidate list capacity per wave (entries beyond are handled by re-scanning)

__global__ void __launch_bounds__(64 * CL_WAVES)
k_clash(int N, int L, const float *__restrict__ xyz, const float *__restrict__ exists,
        const int64_t *__restrict__ rtype, const int64_t *__restrict__ rindex,
        const float *__restrict__ brad, const float *__restrict__ between_radius,
        const float *__restrict__ lower, const float *__restrict__ upper, const int32_t *__restrict__ a2g,
        const float *__restrict__ axes, float tol, float inv_ntot,
        float *__restrict__ per_res, float *__restrict__ dchi) {
    __shared__ int s_list[CL_WAVES][CL_MAXC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * CL_WAVES + wave;
    if (i >= N) return;
    const int b = i / L;
    const int a = lane & 15, slot = lane >> 4;
    const bool own = a < 14;
    const int S = (int)rtype[i];
    const long ri = rindex[i];
    float pa[3] = {0.f, 0.f, 0.f}, ea = 0.f, ra = 0.f;
    if (own) {
#pragma unroll
        for (int k = 0; k < 3; k++) pa[k] = xyz[((size_t)i * 14 + a) * 3 + k];
        ea = exists[(size_t)i * 14 + a];
        ra = ea * between_radius[S * 14 + a];
    }
    // number of side-chain atoms and this residue's weight in the mean
    float nsc = (own && a >= 4) ? ea : 0.f;
    for (int o = 8; o > 0; o >>= 1) nsc += __shfl_xor(nsc, o);
    const float wi = inv_ntot / (nsc + 1e-10f);
    const float cai[3] = {xyz[((size_t)i * 14 + 1) * 3], xyz[((size_t)i * 14 + 1) * 3 + 1], xyz[((size_t)i * 14 + 1) * 3 + 2]};
    const float radi = brad[i];
    const float reach = 3.6f - tol;                 // largest r_a + r_b - tol (S-S)

    float loss_a = 0.f, ga[3] = {0.f, 0.f, 0.f};
    int *list = s_list[wave];
    // partner residues of the same complex, in windows that fit the candidate list
    for (int base = 0; base < L; ) {
        int cnt = 0;
        int jscan = base;
        for (; jscan < L && cnt + 64 <= CL_MAXC; jscan += 64) {
            int jl = jscan + lane;
            bool keep = false;
            if (jl < L) {
                int jg = b * L + jl;
                if (jg != i) {
                    float dx = xyz[((size_t)jg * 14 + 1) * 3] - cai[0], dy = xyz[((size_t)jg * 14 + 1) * 3 + 1] - cai[1],
                          dz = xyz[((size_t)jg * 14 + 1) * 3 + 2] - cai[2];
                    float lim = radi + brad[jg] + reach;
                    keep = (lim > 0.f) && (dx * dx + dy * dy + dz * dz < lim * lim) && (rindex[jg] != ri);
                }
            }
            unsigned long long bal = __ballot(keep);
            if (keep) list[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = b * L + jl;
            cnt += __popcll(bal);
        }
        base = jscan;
        __builtin_amdgcn_wave_barrier();
        for (int c = slot; c < cnt; c += 4) {
            const int jg = list[c];
            const int Sj = (int)rtype[jg];
            const long rj = rindex[jg];
            const bool i_low = ri < rj;
            const bool adjacent = i_low ? (ri + 1 == rj) : (rj + 1 == ri);
            float nscj = 0.f;
#pragma unroll
            for (int bb = 4; bb < 14; bb++) nscj += exists[(size_t)jg * 14 + bb];
            const float wj = inv_ntot / (nscj + 1e-10f);
            if (own && ea != 0.f) {
#pragma unroll
                for (int bb = 0; bb < 14; bb++) {
                    const float eb = exists[(size_t)jg * 14 + bb];
                    bool ok = eb != 0.f && !(a < 4 && bb < 4) && !(a == 5 && bb == 5);
                    if (adjacent) {
                        // peptide bond C(lower) - N(higher)
                        if (i_low ? (a == 2 && bb == 0) : (a == 0 && bb == 2)) ok = false;
                    }
                    if (ok) {
                        const float *pb = xyz + ((size_t)jg * 14 + bb) * 3;
                        float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
                        float d = sqrtf(1e-10f + dx * dx + dy * dy + dz * dz);
                        float rb = eb * between_radius[Sj * 14 + bb];
                        float err = (ra + rb) - tol - d;
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
    // within-residue bounds: stripes of partner atoms b = slot, slot+4, ...
    if (own && ea != 0.f) {
        for (int bb = slot; bb < 14; bb += 4) {
            if (bb == a || (a < 4 && bb < 4)) continue;
            const float eb = exists[(size_t)i * 14 + bb];
            if (eb == 0.f) continue;
            const float *pb = xyz + ((size_t)i * 14 + bb) * 3;
            float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
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
    if (lane == 0) per_res[i] = lres / (nsc + 1e-10f);
    if (dchi) {
        float dk[4] = {0.f, 0.f, 0.f, 0.f};
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
        if (lane < 4) dchi[(size_t)i * 4 + lane] = lane == 0 ? dk[0] : (lane == 1 ? dk[1] : (lane == 2 ? dk[2] : dk[3]));
    }
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

// loss_t = mean_n sum_k (xeff - z)^2 + lamda * mean_n per_res ; then one Adam step on x ; then outputs
__global__ void __launch_bounds__(1024)
k_prox_step(int N, int t, float lamda, float step_size, float bc2s, const float *__restrict__ per_res, const float *__restrict__ dchi,
            const float *__restrict__ chi0, const uint8_t *__restrict__ mask, const float *__restrict__ z,
            float *__restrict__ x, float *__restrict__ m, float *__restrict__ v, float *__restrict__ xeff,
            float *__restrict__ losses, float *__restrict__ traj, float *__restrict__ last) {
    __shared__ float s_part[16];
    const float inv_n = 1.f / (float)N;
    float s = 0.f;
    for (int n = threadIdx.x; n < N; n += 1024) {
        float q = 0.f;
        for (int k = 0; k < 4; k++) {
            float d = xeff[(size_t)n * 4 + k] - z[(size_t)n * 4 + k];
            q += fabsf(d) * fabsf(d);
        }
        s += q + lamda * per_res[n];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tt = 0.f;
        for (int w = 0; w < 16; w++) tt += s_part[w];
        losses[t] = tt * inv_n;
    }
    // torch.optim.Adam defaults: lr 1e-2, betas (0.9, 0.999), eps 1e-8, bias-corrected
    // (step_size = lr / (1 - beta1^t) and bc2s = sqrt(1 - beta2^t) come from the host in double)
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    for (int e = threadIdx.x; e < N * 4; e += 1024) {
        const int n = e >> 2;
        const bool mk = mask[n] != 0;
        float g = 0.f;
        if (mk) g = 2.f * (x[e] - z[e]) * inv_n + lamda * dchi[e];
        float mm = m[e] + (g - m[e]) * (1.f - b1);             // exp_avg.lerp_(grad, 1 - beta1)
        float vv = v[e] * b2 + (1.f - b2) * (g * g);
        float denom = sqrtf(vv) / bc2s + eps;
        float xn = x[e] - step_size * (mm / denom);
        m[e] = mm; v[e] = vv; x[e] = xn;
        const float outv = mk ? xn : chi0[e];
        xeff[e] = outv;
        if (traj) traj[(size_t)t * N * 4 + e] = outv;
        if (last) last[e] = outv;
    }
}

pp_status pp_launch_atom14(pp_ctx *c, const float *chi, float *xyz, hipStream_t s) {
    const pp_plan *p = c->plan;
    hipLaunchKernelGGL(k_atom14, dim3((c->N + 63) / 64), dim3(64), 0, s, c->N, c->b.X, c->b.residue_type, c->b.BB_D, chi,
                       p->default_frames, p->atom14_to_group, p->atom14_mask, p->lit_positions, c->b.atom_mask,
                       xyz, c->axes, c->brad);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_clash(pp_ctx *c, const float *xyz, float *per_res, float *dchi, hipStream_t s) {
    const pp_plan *p = c->plan;
    hipLaunchKernelGGL(k_clash, dim3((c->N + CL_WAVES - 1) / CL_WAVES), dim3(64 * CL_WAVES), 0, s, c->N, c->L, xyz,
                       c->b.atom_mask, c->b.residue_type, c->b.residue_index, c->brad, p->between_radius,
                       p->bounds_lower, p->bounds_upper, p->atom14_to_group, c->axes, p->clash_tol,
                       1.0f / (float)c->N, per_res, dchi);
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}

pp_status pp_launch_proximal(pp_ctx *c, const float *chi, float lamda, int nsteps, float *traj, float *chi_last,
                             float *losses, hipStream_t s) {
    pp_status st;
    // clash mask at the incoming angles (optimize.py:5-18)
    if ((st = pp_launch_atom14(c, chi, c->xyz, s)) != PP_OK) return st;
    if ((st = pp_launch_clash(c, c->xyz, c->per_res, nullptr, s)) != PP_OK) return st;
    hipLaunchKernelGGL(k_prox_init, dim3(1), dim3(1024), 0, s, c->N, c->per_res, chi, c->pmask, c->pz, c->px, c->pm,
                       c->pv, c->pxeff);
    for (int t = 0; t < nsteps; t++) {
        if ((st = pp_launch_atom14(c, c->pxeff, c->xyz, s)) != PP_OK) return st;
        if ((st = pp_launch_clash(c, c->xyz, c->per_res, c->dchi, s)) != PP_OK) return st;
        const double bc1 = 1.0 - pow(0.9, (double)(t + 1)), bc2 = 1.0 - pow(0.999, (double)(t + 1));
        hipLaunchKernelGGL(k_prox_step, dim3(1), dim3(1024), 0, s, c->N, t, lamda, (float)(1e-2 / bc1),
                           (float)sqrt(bc2), c->per_res, c->dchi, chi, c->pmask,
                           c->pz, c->px, c->pm, c->pv, c->pxeff, losses, traj, chi_last);
    }
    PP_HIP_CHECK(hipGetLastError());
    return PP_OK;
}
