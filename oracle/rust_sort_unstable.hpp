// rust_sort_unstable.hpp -- the oracle's restatement of Rust's slice::sort_unstable_by (test infrastructure, like the rest of oracle/).
//
// Why this exists: BVHNode::new sorts the triangle indices of every node with `indices.sort_unstable_by(centroid[axis])`
// (/root/reference/src/acceleration/bvh.rs:45-53).  Rust does not SPECIFY the order of equal keys, but the algorithm is deterministic,
// and on meshes with many equal centroids (extruded text, axis-aligned faces) the tie order decides which triangles end up in the
// same leaf -- hence which zero-thickness leaf boxes exist and which triangles the reference never hits.
// The reference's dependency set (Cargo.lock v4, glam 0.30.3, rand 0.9.1: spring 2025) pins the toolchain to Rust >= 1.83, i.e. the
// "ipnsort" implementation that replaced pdqsort in Rust 1.81 (library/core/src/slice/sort/unstable/{mod,quicksort}.rs,
// shared/{pivot,smallsort}.rs); the standard library is not part of /root/reference, so its published algorithm is restated here
// for T = usize (8 bytes: `has_efficient_in_place_swap`, so small_sort_network with threshold 32 and the branchless cyclic Lomuto
// partition):
//   len <= 20: insertion sort | whole slice one run: done (reversed if strictly descending) | else quicksort with limit
//   2*ilog2(len|1): slices <= 32 -> small_sort_network (halves sorted by the optimal 9- / 13-input networks + insertion, stable
//   bidirectional merge); pivot = median of 3 (recursive pseudo-median from 64 elements) at 0, 4*(len/8), 7*(len/8); a pivot equal to
//   the ancestor pivot partitions by <=; partition = swap pivot to front, cyclic Lomuto over the rest, swap pivot to num_lt;
//   heapsort when the limit runs out.
// Evidence that the restatement is the reference's sort: with it the oracle's replay of the reference stream matches the
// reference's own committed render docs/semesterbild.png bit for bit, all 480 000 pixels (std::stable_sort instead: whole letter
// faces differ); tests/test_oracle_golden.py.  tests/test_bvh.py lists which branches of the algorithm that mesh's build goes
// through (g_paths below) -- those are pinned by the reference's render, the others only by this restatement.
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>
namespace rustsort {
// Which branches of the algorithm a build went through (test infrastructure: tests/test_bvh.py reports which of them the BVH of the
// reference's own golden-pinned mesh exercises, i.e. which are held by docs/semesterbild.png and which only by this restatement).
enum Path { P_CALLS, P_INSERTION_ONLY, P_RUN_ASCENDING, P_RUN_DESCENDING, P_QUICKSORT, P_SMALL_NO_MERGE, P_SORT9, P_SORT13, P_MERGE, P_MERGE_ODD,
            P_MEDIAN3, P_MEDIAN3_REC, P_PARTITION_LT, P_PARTITION_LE_ANCESTOR, P_HEAPSORT, P_COUNT };
inline uint64_t g_paths[P_COUNT] = {};
inline void hit(Path p) { ++g_paths[p]; }
template <class T, class F> void insertion_sort_shift_left(T* v, size_t len, size_t offset, F& is_less) {
    for (size_t i = offset; i < len; ++i) {                       // insert_tail(v[..=i])
        if (is_less(v[i], v[i - 1])) {
            T tmp = v[i]; size_t j = i;
            do { v[j] = v[j - 1]; --j; } while (j > 0 && is_less(tmp, v[j - 1]));
            v[j] = tmp;
        }
    }
}
template <class T, class F> inline void swap_if_less(T* v, size_t a, size_t b, F& is_less) { if (is_less(v[b], v[a])) std::swap(v[a], v[b]); }
template <class T, class F> void sort9_optimal(T* v, F& is_less) {
    static const uint8_t P[25][2] = {{0,3},{1,7},{2,5},{4,8},{0,7},{2,4},{3,8},{5,6},{0,2},{1,3},{4,5},{7,8},{1,4},{3,6},{5,7},{0,1},{2,4},{3,5},{6,8},{2,3},{4,5},{6,7},{1,2},{3,4},{5,6}};
    for (auto& p : P) swap_if_less(v, p[0], p[1], is_less);
}
template <class T, class F> void sort13_optimal(T* v, F& is_less) {
    static const uint8_t P[45][2] = {{0,12},{1,10},{2,9},{3,7},{5,11},{6,8},{1,6},{2,3},{4,11},{7,9},{8,10},{0,4},{1,2},{3,6},{7,8},{9,10},{11,12},{4,6},{5,9},{8,11},{10,12},
        {0,5},{3,8},{4,7},{6,11},{9,10},{0,1},{2,5},{6,9},{7,8},{10,11},{1,3},{2,4},{5,6},{9,10},{1,2},{3,4},{5,7},{6,8},{2,3},{4,5},{6,7},{8,9},{3,4},{5,6}};
    for (auto& p : P) swap_if_less(v, p[0], p[1], is_less);
}
template <class T, class F> void bidirectional_merge(const T* src, size_t len, T* dst, F& is_less) {
    const size_t half = len / 2;
    const T *left = src, *right = src + half, *left_rev = src + half - 1, *right_rev = src + len - 1;
    T *d = dst, *d_rev = dst + len - 1;
    for (size_t k = 0; k < half; ++k) {
        { const bool is_l = !is_less(*right, *left); *d++ = is_l ? *left : *right; right += !is_l; left += is_l; }                    // merge_up
        { const bool is_l = !is_less(*right_rev, *left_rev); *d_rev-- = is_l ? *right_rev : *left_rev; right_rev -= is_l; left_rev -= !is_l; }   // merge_down
    }
    if (len % 2 != 0) { hit(P_MERGE_ODD); const bool left_nonempty = left < left_rev + 1; *d = left_nonempty ? *left : *right; }
}
template <class T, class F> void small_sort_network(T* v, size_t len, F& is_less) {
    if (len < 2) return;
    const size_t half = len / 2; const bool no_merge = len < 18;
    T* region = v; size_t rlen = no_merge ? len : half;
    for (;;) {
        size_t presorted = 1;
        if (rlen >= 13) { hit(P_SORT13); sort13_optimal(region, is_less); presorted = 13; }
        else if (rlen >= 9) { hit(P_SORT9); sort9_optimal(region, is_less); presorted = 9; }
        insertion_sort_shift_left(region, rlen, presorted, is_less);
        if (no_merge) { hit(P_SMALL_NO_MERGE); return; }
        if (region != v) break;
        region = v + half; rlen = len - half;
    }
    T scratch[32];
    hit(P_MERGE);
    bidirectional_merge(v, len, scratch, is_less);
    for (size_t i = 0; i < len; ++i) v[i] = scratch[i];
}
template <class T, class F> const T* median3(const T* a, const T* b, const T* c, F& is_less) {
    const bool x = is_less(*a, *b), y = is_less(*a, *c);
    if (x == y) { const bool z = is_less(*b, *c); return (z ^ x) ? c : b; }
    return a;
}
template <class T, class F> const T* median3_rec(const T* a, const T* b, const T* c, size_t n, F& is_less) {
    if (n * 8 >= 64) {
        const size_t n8 = n / 8;
        a = median3_rec(a, a + n8 * 4, a + n8 * 7, n8, is_less);
        b = median3_rec(b, b + n8 * 4, b + n8 * 7, n8, is_less);
        c = median3_rec(c, c + n8 * 4, c + n8 * 7, n8, is_less);
    }
    return median3(a, b, c, is_less);
}
template <class T, class F> size_t choose_pivot(const T* v, size_t len, F& is_less) {
    const size_t len_div_8 = len / 8;
    const T *a = v, *b = v + len_div_8 * 4, *c = v + len_div_8 * 7;
    hit(len < 64 ? P_MEDIAN3 : P_MEDIAN3_REC);
    return (size_t)((len < 64 ? median3(a, b, c, is_less) : median3_rec(a, b, c, len_div_8, is_less)) - v);
}
// partition_lomuto_branchless_cyclic over v[0..len) (the slice WITHOUT the pivot)
template <class T, class F> size_t lomuto_cyclic(T* v, size_t len, const T& pivot, F& is_less) {
    if (len == 0) return 0;
    const T gap_value = v[0]; size_t gap_pos = 0, num_lt = 0;
    for (size_t right = 1; right < len; ++right) {
        const bool lt = is_less(v[right], pivot);
        v[gap_pos] = v[num_lt]; v[num_lt] = v[right]; gap_pos = right; num_lt += lt;
    }
    { const bool lt = is_less(gap_value, pivot); v[gap_pos] = v[num_lt]; v[num_lt] = gap_value; num_lt += lt; }
    return num_lt;
}
template <class T, class F> size_t partition(T* v, size_t len, size_t pivot_pos, F& is_less) {
    if (len == 0) return 0;
    std::swap(v[0], v[pivot_pos]);
    const T pivot = v[0];
    const size_t num_lt = lomuto_cyclic(v + 1, len - 1, pivot, is_less);
    std::swap(v[0], v[num_lt]);
    return num_lt;
}
template <class T, class F> void heapsort(T* v, size_t len, F& is_less) {
    auto sift_down = [&](T* s, size_t n, size_t node) {
        for (;;) { size_t child = 2 * node + 1; if (child >= n) break; if (child + 1 < n) child += is_less(s[child], s[child + 1]) ? 1 : 0;
                   if (!is_less(s[node], s[child])) break;
                   std::swap(s[node], s[child]); node = child; }
    };
    for (size_t i = len + len / 2; i-- > 0;) {
        size_t sift_idx; if (i >= len) sift_idx = i - len; else { std::swap(v[0], v[i]); sift_idx = 0; }
        sift_down(v, i < len ? i : len, sift_idx);
    }
}
template <class T, class F> void quicksort(T* v, size_t len, const T* ancestor_pivot, uint32_t limit, F& is_less) {
    for (;;) {
        if (len <= 32) { small_sort_network(v, len, is_less); return; }
        if (limit == 0) { hit(P_HEAPSORT); heapsort(v, len, is_less); return; }
        --limit;
        const size_t pivot_pos = choose_pivot(v, len, is_less);
        if (ancestor_pivot && !is_less(*ancestor_pivot, v[pivot_pos])) {
            hit(P_PARTITION_LE_ANCESTOR);
            auto le = [&](const T& a, const T& b) { return !is_less(b, a); };
            const size_t num_le = partition(v, len, pivot_pos, le);
            v += num_le + 1; len -= num_le + 1; ancestor_pivot = nullptr;
            continue;
        }
        hit(P_PARTITION_LT);
        const size_t num_lt = partition(v, len, pivot_pos, is_less);
        quicksort(v, num_lt, ancestor_pivot, limit, is_less);
        ancestor_pivot = v + num_lt;
        v += num_lt + 1; len -= num_lt + 1;
    }
}
template <class T, class F> void sort_unstable_by(T* v, size_t len, F is_less) {
    hit(P_CALLS);
    if (len < 2) return;
    if (len <= 20) { hit(P_INSERTION_ONLY); insertion_sort_shift_left(v, len, 1, is_less); return; }
    // ipnsort: an existing run over the whole slice?
    size_t run = 2; const bool desc = is_less(v[1], v[0]);
    if (desc) while (run < len && is_less(v[run], v[run - 1])) ++run;
    else      while (run < len && !is_less(v[run], v[run - 1])) ++run;
    if (run == len) { hit(desc ? P_RUN_DESCENDING : P_RUN_ASCENDING); if (desc) for (size_t i = 0, j = len - 1; i < j; ++i, --j) std::swap(v[i], v[j]); return; }
    uint32_t lg = 0; for (size_t x = len | 1; x > 1; x >>= 1) ++lg;
    hit(P_QUICKSORT);
    quicksort(v, len, (const T*)nullptr, 2u * lg, is_less);
}
}  // namespace rustsort
