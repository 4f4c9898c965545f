// The neighbour choice of the reference's CPU path on equal distances.
//
// encoder.py:105-118 takes `torch.topk(D_adjust, K, dim=-1, largest=False)`.  On CPU, ATen (TopKImpl.h, topk_impl_loop)
// fills a queue of (value, index) pairs per row and runs, with a comparator that looks at the VALUES only,
//     k * 64 <= n :  std::partial_sort(queue, queue + k, end)
//     otherwise   :  std::nth_element(queue, queue + k - 1, end);  std::sort(queue, queue + k - 1)
// and returns the first k pairs.  Which of several equal values makes it into the list (and in which order) is therefore a
// property of libstdc++'s introselect / introsort / heap code on that exact sequence.  This header restates those algorithms
// (GCC's bits/stl_algo.h and bits/stl_heap.h: __introselect, __unguarded_partition_pivot, __move_median_to_first,
// __insertion_sort, __introsort_loop, __final_insertion_sort, __heap_select, __adjust_heap, __push_heap, __sort_heap) for a
// plain array of pairs so that ONE lane of the kNN kernel can run them on a row held in LDS.  The same code compiles for the
// host: `pp_topk_aten_host` (pp_api.hip) is checked against torch.topk on tie-heavy rows by the CPU tests, and against
// std::nth_element / std::sort / std::partial_sort themselves by tests/test_topk_aten.py.
//
// PROVENANCE / LICENCE NOTE.  The control flow below follows, step for step, the algorithms of the GNU C++ Library (libstdc++),
// files bits/stl_algo.h and bits/stl_heap.h, Copyright (C) 2001-2023 Free Software Foundation, Inc. (and, for the portions
// derived from the SGI STL, Copyright (c) 1994 Hewlett-Packard Company and Copyright (c) 1996 Silicon Graphics Computer Systems,
// Inc.), distributed under the GNU General Public License version 3 or later WITH the GCC Runtime Library Exception 3.1, which
// permits use of the library's algorithms in programs of any licence.  No libstdc++ source text is reproduced here: the
// functions are rewritten for a fixed element type and a fixed comparator, for host and device, because bit-exact agreement
// with torch.topk on ties REQUIRES the same sequence of comparisons and moves.  ATen's TopKImpl.h (the call pattern quoted
// above) is part of PyTorch, BSD-3-Clause.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define PP_HD __host__ __device__ __forceinline__
#define PP_HD_NOINLINE static __host__ __device__ __attribute__((noinline))
#else
#define PP_HD inline
#define PP_HD_NOINLINE static inline
#endif

struct __attribute__((aligned(8))) pp_tk_pair {
    float v;
    int32_t i;
};

// ATen's "smallest first, NaN last" comparator: ((!isnan(x) && isnan(y)) || x < y)
PP_HD bool pp_tk_less(const pp_tk_pair &x, const pp_tk_pair &y) {
    return ((x.v == x.v) && (y.v != y.v)) || (x.v < y.v);
}
PP_HD void pp_tk_swap(pp_tk_pair *a, pp_tk_pair *b) {
    pp_tk_pair t = *a;
    *a = *b;
    *b = t;
}
PP_HD int pp_tk_lg(int n) {      // std::__lg: floor(log2(n)), n > 0
    int r = 0;
    while (n > 1) { n >>= 1; r++; }
    return r;
}

// ---- heap (bits/stl_heap.h) --------------------------------------------------------------------------------------
PP_HD void pp_tk_push_heap(pp_tk_pair *first, int hole, int top, pp_tk_pair value) {
    int parent = (hole - 1) / 2;
    while (hole > top && pp_tk_less(first[parent], value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}
PP_HD void pp_tk_adjust_heap(pp_tk_pair *first, int hole, int len, pp_tk_pair value) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (pp_tk_less(first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    pp_tk_push_heap(first, hole, top, value);
}
PP_HD void pp_tk_make_heap(pp_tk_pair *first, pp_tk_pair *last) {
    const int len = (int)(last - first);
    if (len < 2) return;
    int parent = (len - 2) / 2;
    while (true) {
        pp_tk_pair value = first[parent];
        pp_tk_adjust_heap(first, parent, len, value);
        if (parent == 0) return;
        parent--;
    }
}
PP_HD void pp_tk_pop_heap(pp_tk_pair *first, pp_tk_pair *last, pp_tk_pair *result) {
    pp_tk_pair value = *result;
    *result = *first;
    pp_tk_adjust_heap(first, 0, (int)(last - first), value);
}
PP_HD void pp_tk_heap_select(pp_tk_pair *first, pp_tk_pair *middle, pp_tk_pair *last) {
    pp_tk_make_heap(first, middle);
    for (pp_tk_pair *i = middle; i < last; ++i)
        if (pp_tk_less(*i, *first)) pp_tk_pop_heap(first, middle, i);
}
PP_HD void pp_tk_sort_heap(pp_tk_pair *first, pp_tk_pair *last) {
    while (last - first > 1) {
        --last;
        pp_tk_pop_heap(first, last, last);
    }
}
PP_HD void pp_tk_partial_sort(pp_tk_pair *first, pp_tk_pair *middle, pp_tk_pair *last) {
    pp_tk_heap_select(first, middle, last);
    pp_tk_sort_heap(first, middle);
}

// ---- partition / insertion sort (bits/stl_algo.h) ----------------------------------------------------------------------
PP_HD void pp_tk_median_to_first(pp_tk_pair *result, pp_tk_pair *a, pp_tk_pair *b, pp_tk_pair *c) {
    if (pp_tk_less(*a, *b)) {
        if (pp_tk_less(*b, *c)) pp_tk_swap(result, b);
        else if (pp_tk_less(*a, *c)) pp_tk_swap(result, c);
        else pp_tk_swap(result, a);
    } else if (pp_tk_less(*a, *c)) pp_tk_swap(result, a);
    else if (pp_tk_less(*b, *c)) pp_tk_swap(result, c);
    else pp_tk_swap(result, b);
}
PP_HD pp_tk_pair *pp_tk_partition_pivot(pp_tk_pair *first, pp_tk_pair *last) {
    pp_tk_pair *mid = first + (last - first) / 2;
    pp_tk_median_to_first(first, first + 1, mid, last - 1);
    // __unguarded_partition(first + 1, last, pivot = first)
    pp_tk_pair *lo = first + 1, *hi = last;
    const pp_tk_pair *pivot = first;
    while (true) {
        while (pp_tk_less(*lo, *pivot)) ++lo;
        --hi;
        while (pp_tk_less(*pivot, *hi)) --hi;
        if (!(lo < hi)) return lo;
        pp_tk_swap(lo, hi);
        ++lo;
    }
}
PP_HD void pp_tk_unguarded_linear_insert(pp_tk_pair *last) {
    pp_tk_pair val = *last;
    pp_tk_pair *next = last - 1;
    while (pp_tk_less(val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}
PP_HD void pp_tk_insertion_sort(pp_tk_pair *first, pp_tk_pair *last) {
    if (first == last) return;
    for (pp_tk_pair *i = first + 1; i != last; ++i) {
        if (pp_tk_less(*i, *first)) {
            pp_tk_pair val = *i;
            for (pp_tk_pair *p = i; p != first; --p) *p = *(p - 1);      // std::move_backward(first, i, i + 1)
            *first = val;
        } else
            pp_tk_unguarded_linear_insert(i);
    }
}

// ---- the same partition as a DATA-PARALLEL formula (what the kNN kernel's 256 threads evaluate together) ------------------
// __unguarded_partition(first + 1, last, pivot = *first) only ever exchanges disjoint pairs, and which pairs is a closed form
// of the ORIGINAL sequence: let A = the positions i in [first + 1, last), ascending, with !(a[i] < pivot) (where the upward
// scan can stop) and B = the positions, descending, with !(pivot < a[i]) (where the downward scan can stop), followed by
// `first` itself (the pivot: the scan's sentinel).  Between two exchanges both scans only cross untouched elements, so the
// t-th exchange is (A[t], B[t]); it happens while A[t] < B[t]; with T = the number of exchanges the function returns
// min(A[T], B[T - 1]) (the upward scan stops at the latest on the element the last exchange put at B[T - 1]; B[-1] = last).
// This sequential statement of the formula is what tests/test_topk_aten.py holds against std::nth_element; the kernel computes
// A and B by prefix sums, T by a reduction and does the exchanges in parallel (pp_prepare.hip).
PP_HD pp_tk_pair *pp_tk_partition_pivot_lists(pp_tk_pair *first, pp_tk_pair *last, int *A, int *B) {
    pp_tk_pair *mid = first + (last - first) / 2;
    pp_tk_median_to_first(first, first + 1, mid, last - 1);
    const pp_tk_pair pivot = *first;
    const int m = (int)(last - first);
    int nA = 0, nB = 0;
    for (int i = 1; i < m; i++)
        if (!pp_tk_less(first[i], pivot)) A[nA++] = i;
    for (int i = m - 1; i >= 1; i--)
        if (!pp_tk_less(pivot, first[i])) B[nB++] = i;
    B[nB++] = 0;                                   // the pivot position: where the downward scan stops at the latest
    int T = 0;
    while (T < nA && T < nB && A[T] < B[T]) T++;
    for (int t = 0; t < T; t++) pp_tk_swap(first + A[t], first + B[t]);
    const int a_next = T < nA ? A[T] : m, b_prev = T > 0 ? B[T - 1] : m;
    return first + (a_next < b_prev ? a_next : b_prev);
}
// std::nth_element with that partition (scratch: two int arrays of last - first entries)
PP_HD void pp_tk_nth_element_lists(pp_tk_pair *first, pp_tk_pair *nth, pp_tk_pair *last, int *A, int *B) {
    if (first == last || nth == last) return;
    int depth = pp_tk_lg((int)(last - first)) * 2;
    while (last - first > 3) {
        if (depth == 0) {
            pp_tk_heap_select(first, nth + 1, last);
            pp_tk_swap(first, nth);
            return;
        }
        --depth;
        pp_tk_pair *cut = pp_tk_partition_pivot_lists(first, last, A, B);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    pp_tk_insertion_sort(first, last);
}

// std::nth_element(first, nth, last)
PP_HD void pp_tk_nth_element(pp_tk_pair *first, pp_tk_pair *nth, pp_tk_pair *last) {
    if (first == last || nth == last) return;
    int depth = pp_tk_lg((int)(last - first)) * 2;
    while (last - first > 3) {
        if (depth == 0) {
            pp_tk_heap_select(first, nth + 1, last);
            pp_tk_swap(first, nth);
            return;
        }
        --depth;
        pp_tk_pair *cut = pp_tk_partition_pivot(first, last);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    pp_tk_insertion_sort(first, last);
}

// std::sort(first, last)
PP_HD_NOINLINE void pp_tk_sort(pp_tk_pair *first, pp_tk_pair *last) {
    if (first == last) return;
    // __introsort_loop: the recursion on (cut, last) becomes an explicit stack; the ranges are disjoint and the
    // comparator only sees values inside a range, so the order in which they are finished does not matter
    struct Range { int f, l, d; } stack[64];
    int sp = 0;
    stack[sp++] = {0, (int)(last - first), pp_tk_lg((int)(last - first)) * 2};
    while (sp > 0) {
        Range r = stack[--sp];
        pp_tk_pair *f = first + r.f, *l = first + r.l;
        int d = r.d;
        while (l - f > 16) {
            if (d == 0) {
                pp_tk_partial_sort(f, l, l);
                break;
            }
            --d;
            pp_tk_pair *cut = pp_tk_partition_pivot(f, l);
            if (sp < 64) stack[sp++] = {(int)(cut - first), (int)(l - first), d};
            l = cut;
        }
    }
    // __final_insertion_sort
    if (last - first > 16) {
        pp_tk_insertion_sort(first, first + 16);
        for (pp_tk_pair *i = first + 16; i != last; ++i) pp_tk_unguarded_linear_insert(i);
    } else
        pp_tk_insertion_sort(first, last);
}

// ATen's topk_impl_loop for largest = false, sorted = true on a queue of n pairs (queue[j] = (value_j, j) on entry);
// the answer is queue[0 .. k-1] on return.
PP_HD_NOINLINE void pp_tk_topk_smallest(pp_tk_pair *queue, int n, int k) {
    if (k <= 0 || n <= 0) return;
    if (k > n) k = n;
    if ((long)k * 64 <= (long)n) {
        pp_tk_partial_sort(queue, queue + k, queue + n);
    } else {
        pp_tk_nth_element(queue, queue + k - 1, queue + n);
        pp_tk_sort(queue, queue + k - 1);
    }
}
