// Test harness (g++): the restated selection of csrc/pp_topk_aten.h next to the real libstdc++ algorithms ATen's CPU topk
// calls (TopKImpl.h: std::partial_sort, or std::nth_element + std::sort of the first k - 1, value-only comparator).
#include "../../packppi_amd/csrc/pp_topk_aten.h"
#include <algorithm>
#include <cmath>
#include <utility>
#include <vector>

using elem_t = std::pair<double, int64_t>;
static bool less_nan_last(const elem_t &x, const elem_t &y) {
    return ((!std::isnan(x.first) && std::isnan(y.first)) || (x.first < y.first));
}
static std::vector<elem_t> fill(const float *v, int n) {
    std::vector<elem_t> q(n);
    for (int j = 0; j < n; j++) { q[j].first = v[j]; q[j].second = j; }
    return q;
}
static std::vector<pp_tk_pair> fill_mine(const float *v, int n) {
    std::vector<pp_tk_pair> q(n);
    for (int j = 0; j < n; j++) { q[j].v = v[j]; q[j].i = j; }
    return q;
}

extern "C" void mine_topk(const float *v, int n, int k, int *out) {
    auto q = fill_mine(v, n);
    pp_tk_topk_smallest(q.data(), n, k);
    for (int j = 0; j < k; j++) out[j] = q[j].i;
}
extern "C" void std_topk(const float *v, int n, int k, int *out) {
    auto q = fill(v, n);
    if ((long)k * 64 <= n) std::partial_sort(q.begin(), q.begin() + k, q.end(), less_nan_last);
    else {
        std::nth_element(q.begin(), q.begin() + k - 1, q.end(), less_nan_last);
        std::sort(q.begin(), q.begin() + k - 1, less_nan_last);
    }
    for (int j = 0; j < k; j++) out[j] = (int)q[j].second;
}
// the pieces on their own, whole permutation compared: which = 0 sort, 1 nth_element(k - 1), 2 partial_sort(k)
extern "C" void mine_piece(int which, const float *v, int n, int k, int *out) {
    auto q = fill_mine(v, n);
    if (which == 0) pp_tk_sort(q.data(), q.data() + n);
    else if (which == 1) pp_tk_nth_element(q.data(), q.data() + k - 1, q.data() + n);
    else if (which == 3) {      // nth_element through the data-parallel partition formula (the kernel's form)
        std::vector<int> A(n + 1), B(n + 1);
        pp_tk_nth_element_lists(q.data(), q.data() + k - 1, q.data() + n, A.data(), B.data());
    }
    else pp_tk_partial_sort(q.data(), q.data() + k, q.data() + n);
    for (int j = 0; j < n; j++) out[j] = q[j].i;
}
extern "C" void std_piece(int which, const float *v, int n, int k, int *out) {
    auto q = fill(v, n);
    if (which == 0) std::sort(q.begin(), q.end(), less_nan_last);
    else if (which == 1 || which == 3) std::nth_element(q.begin(), q.begin() + k - 1, q.end(), less_nan_last);
    else std::partial_sort(q.begin(), q.begin() + k, q.end(), less_nan_last);
    for (int j = 0; j < n; j++) out[j] = (int)q[j].second;
}
