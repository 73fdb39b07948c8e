// nm_select.h -- which M-N elements of one N:M group get pruned, with the reference's tie order.
//
// The reference prunes with torch.topk(|v|, k=M-N, largest=False) (bfp_ops.py:85).  On the CPU path
// that is libstdc++ std::nth_element(first, first+k-1, last) over (|v|, index) pairs followed by
// "take the first k slots" -- so among equal magnitudes the pruned ones are wherever introselect's
// median-of-3 partitioning, heap-select fallback and final insertion sort leave them.
// This header replays that sequence of comparisons and moves on a small array that may live in
// LDS (device, one column per thread) or in host memory; it is shared by the device kernels for
// general M and by the host code that builds the 729-entry table for M = 4.
//
// Element encoding: uint64  (magnitude_key << 8) | index,  index < 64; only the key part is compared.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BFPQ_HD __host__ __device__ __forceinline__
#else
#define BFPQ_HD inline
#endif

namespace bfpq {

// A strided view: element i of this thread's private array is base[i * stride].
struct KvView {
    uint64_t* base;
    int stride;
    BFPQ_HD uint64_t get(int i) const { return base[(int64_t)i * stride]; }
    BFPQ_HD void set(int i, uint64_t v) const { base[(int64_t)i * stride] = v; }
    BFPQ_HD void swap(int i, int j) const { uint64_t a = get(i), b = get(j); set(i, b); set(j, a); }
};

BFPQ_HD bool kv_less(uint64_t a, uint64_t b) { return (a >> 8) < (b >> 8); }

// sift `value` down from `hole` in the max-heap a[first .. first+len), then back up (libstdc++ __adjust_heap)
BFPQ_HD void nm_adjust_heap(const KvView& a, int first, int hole, int len, uint64_t value)
{
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (kv_less(a.get(first + child), a.get(first + child - 1))) child--;
        a.set(first + hole, a.get(first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a.set(first + hole, a.get(first + child - 1));
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && kv_less(a.get(first + parent), value)) {
        a.set(first + hole, a.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a.set(first + hole, value);
}

// smallest (middle-first) elements of [first,last) end up, as a max-heap, in [first,middle)
BFPQ_HD void nm_heap_select(const KvView& a, int first, int middle, int last)
{
    const int len = middle - first;
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; parent--) {
            nm_adjust_heap(a, first, parent, len, a.get(first + parent));
            if (parent == 0) break;
        }
    }
    for (int i = middle; i < last; i++) {
        if (kv_less(a.get(i), a.get(first))) {
            uint64_t v = a.get(i);
            a.set(i, a.get(first));
            nm_adjust_heap(a, first, 0, len, v);
        }
    }
}

BFPQ_HD void nm_insertion_sort(const KvView& a, int first, int last)
{
    for (int i = first + 1; i < last; i++) {
        uint64_t v = a.get(i);
        if (kv_less(v, a.get(first))) {
            for (int j = i; j > first; j--) a.set(j, a.get(j - 1));
            a.set(first, v);
        } else {
            int j = i;
            while (kv_less(v, a.get(j - 1))) { a.set(j, a.get(j - 1)); j--; }
            a.set(j, v);
        }
    }
}

// After the call the k = nth+1 smallest-by-key elements (reference tie order) sit in a[0..nth].
BFPQ_HD void nm_nth_element(const KvView& a, int n, int nth)
{
    if (n == 0 || nth >= n) return;
    int first = 0, last = n;
    int depth = 0;
    for (int t = n; t > 1; t >>= 1) depth += 2;          // 2 * floor(log2 n)
    while (last - first > 3) {
        if (depth == 0) {
            nm_heap_select(a, first, nth + 1, last);
            a.swap(first, nth);
            return;
        }
        depth--;
        // median of a[first+1], a[mid], a[last-1] goes to a[first]
        const int x = first + 1, y = first + (last - first) / 2, z = last - 1;
        const uint64_t vx = a.get(x), vy = a.get(y), vz = a.get(z);
        int med;
        if (kv_less(vx, vy)) med = kv_less(vy, vz) ? y : (kv_less(vx, vz) ? z : x);
        else med = kv_less(vx, vz) ? x : (kv_less(vy, vz) ? z : y);
        a.swap(first, med);
        // Hoare partition of [first+1, last) around the pivot now at a[first]
        const uint64_t pivot = a.get(first);
        int lo = first + 1, hi = last;
        for (;;) {
            while (kv_less(a.get(lo), pivot)) lo++;
            hi--;
            while (kv_less(pivot, a.get(hi))) hi--;
            if (!(lo < hi)) break;
            a.swap(lo, hi);
            lo++;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
    nm_insertion_sort(a, first, last);
}

// Prune mask (bit i set = element i zeroed) for "keep N of M"; keys[i] already in a (index order).
BFPQ_HD uint64_t nm_prune_mask(const KvView& a, int N, int M)
{
    const int k = M - N;
    if (k <= 0) return 0;
    nm_nth_element(a, M, k - 1);
    uint64_t mask = 0;
    for (int i = 0; i < k; i++) mask |= 1ull << (a.get(i) & 0xff);
    return mask;
}

}  // namespace bfpq
