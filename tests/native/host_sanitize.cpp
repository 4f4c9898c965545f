// CPU-built sanitizer harness (g++ -fsanitize=address,undefined; tests/test_sanitize.py): the HOST-side C++ of the library that can be
// compiled without HIP -- the restated libstdc++ selection of csrc/pp_topk_aten.h (what pp_topk_aten_host and the kNN kernel's tie
// path run) and the checkpoint rewrites of csrc/pp_rebalance.h (what pp_rebalance_weights_host / pp_plan_create run) -- driven over
// tie-heavy rows, degenerate sizes and hostile weight vectors.  Exit code 0 and an empty sanitizer report = pass.  Never run on the GPU.
#include "../../packppi_amd/csrc/pp_topk_aten.h"
#include "../../packppi_amd/csrc/pp_rebalance.h"

#include <cstdint>
#include <cstdio>
#include <limits>
#include <random>
#include <vector>

static int topk_rows(unsigned seed, int rows) {
    std::mt19937 rng(seed);
    int checked = 0;
    for (int t = 0; t < rows; t++) {
        const int n = t % 7 == 0 ? 1 + (int)(rng() % 4) : (t % 3 ? 1 + (int)(rng() % 3000) : 2048 + (int)(rng() % 4000));
        const int k = std::min(n, t % 5 ? 32 : 1 + (int)(rng() % 64));
        const int levels = (int[]){1, 2, 5, 20, 1000, 100000}[rng() % 6];
        std::vector<pp_tk_pair> q((size_t)n);
        for (int j = 0; j < n; j++) {
            q[j].v = 0.37f * (float)(rng() % levels);
            q[j].i = j;
            if (t % 13 == 0 && rng() % 50 == 0) q[j].v = std::numeric_limits<float>::quiet_NaN();
            if (t % 17 == 0 && rng() % 40 == 0) q[j].v = std::numeric_limits<float>::infinity();
        }
        if (t % 11 == 0) std::sort(q.begin(), q.end(), [](const pp_tk_pair &a, const pp_tk_pair &b) { return a.v < b.v; });
        std::vector<pp_tk_pair> a = q, b = q, c = q;
        pp_tk_topk_smallest(a.data(), n, k);
        // the pieces on their own, including the data-parallel partition lists of the kNN kernel (scratch of n + 1 entries each)
        pp_tk_sort(b.data(), b.data() + n);
        std::vector<int> A((size_t)n + 1), B((size_t)n + 1);
        pp_tk_nth_element_lists(c.data(), c.data() + k - 1, c.data() + n, A.data(), B.data());
        std::vector<pp_tk_pair> d = q;
        pp_tk_partial_sort(d.data(), d.data() + k, d.data() + n);
        // every output is a permutation of the input indices
        std::vector<char> seen((size_t)n, 0);
        for (int j = 0; j < n; j++) {
            if (c[j].i < 0 || c[j].i >= n || seen[c[j].i]) { std::fprintf(stderr, "not a permutation (row %d)\n", t); return -1; }
            seen[c[j].i] = 1;
        }
        checked++;
    }
    return checked;
}

static int rebalance_cases(unsigned seed) {
    const WeightOff off = pp_weight_offsets();
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    int chains = 0;
    for (int variant = 0; variant < 7; variant++) {
        std::vector<float> w(off.total);
        for (auto &x : w) x = 0.05f * nd(rng);
        const LayerOff &L = off.layer[variant % 3];
        auto scale = [&](size_t at, size_t n, float s) { for (size_t i = 0; i < n; i++) w[at + i] *= s; };
        switch (variant) {
        case 0: break;                                                                       // balanced: nothing to do
        case 1: scale(L.em_in_w, 128 * 456, 1e-3f); scale(L.em_mid_w, 128 * 128, 1e3f); break;   // tiny hidden operands
        case 2: scale(L.ed_in_w, 512 * 128, 3e5f); scale(L.ed_out_w, 128 * 512, 1.f / 3e5f); break;  // huge hidden layer
        case 3: scale(L.nm_in_w, 128 * 456, 1e-4f); scale(L.nm_in_w, 456, 3e7f); break;       // small median, one enormous row
        case 4: for (size_t i = 0; i < 128 * 456; i++) w[L.em_in_w + i] = 0.f; break;          // an all-zero producer
        case 5: w[L.em_mid_w + 5] = 6.0e4f; scale(L.em_in_w, 128 * 456, 1e-5f); break;        // a consumer entry at the f16 edge
        case 6: scale(L.ed_in_w, 512 * 128, 1e-30f); break;                                   // denormal producer
        }
        if (variant == 3 || variant == 5) {          // small LayerNorm gains as well: operand scales, then the chains
            for (int f = 0; f < 128; f++) { w[L.norm_g[2] + f] = 1e-3f * (1.f + f % 7); w[L.norm_b[2] + f] = 2e-4f * (f % 5); }
            std::vector<float> plain, packed;
            LnScales sc;
            chains += rewrite_checkpoint(w.data(), off, plain, packed, sc);
            for (size_t i = 0; i < w.size(); i++)
                if ((!(std::fabs(plain[i]) < 65504.f) || !(std::fabs(packed[i]) < 65504.f)) && std::fabs(w[i]) < 65504.f && variant != 5) {
                    std::fprintf(stderr, "variant %d: rewritten weight %zu left the f16 range\n", variant, i);
                    return -1;
                }
        }
        const std::vector<float> before = w;
        chains += rebalance_relu_chains(w.data(), off);
        for (size_t i = 0; i < w.size(); i++)
            if (!(std::fabs(w[i]) < 65504.f) && std::fabs(before[i]) < 65504.f) {
                std::fprintf(stderr, "variant %d: weight %zu left the f16 range (%g -> %g)\n", variant, i, (double)before[i], (double)w[i]);
                return -1;
            }
    }
    return chains;
}

int main() {
    const int rows = topk_rows(7u, 1200);
    if (rows < 0) return 1;
    const int chains = rebalance_cases(3u);
    if (chains < 0) return 1;
    std::printf("ok: %d top-k rows, %d rebalanced chains\n", rows, chains);
    return 0;
}
