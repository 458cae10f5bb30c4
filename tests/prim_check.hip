// prim_check.hip -- stand-alone check of smvp-toolkit_amd/csrc/smvp_prim.h (the hand-written radix sort and prefix sums behind the
// device-side converters and the plan builders) against std::stable_sort / running sums on the host.  Built and run by
// tests/test_gpu_parity.py::test_device_primitives (hipcc on the GPU box); prints "prim ok" or the first mismatch.
#include "smvp_prim.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static int g_bad = 0;

template <class K>
static void check_sort(size_t n, unsigned begin_bit, unsigned end_bit, unsigned key_spread_bits, unsigned seed)
{
    std::mt19937_64 rng(seed);
    std::vector<K> keys(n);
    std::vector<unsigned> vals(n);
    const unsigned long long spread = key_spread_bits >= 64 ? ~0ull : ((1ull << key_spread_bits) - 1);
    for (size_t i = 0; i < n; ++i) {
        keys[i] = (K)(rng() & spread);
        vals[i] = (unsigned)i;  // the original position: a stable sort keeps equal keys in this order
    }
    K *k0, *k1;
    unsigned *v0, *v1;
    CK(hipMalloc(&k0, (n + 1) * sizeof(K))); CK(hipMalloc(&k1, (n + 1) * sizeof(K)));
    CK(hipMalloc(&v0, (n + 1) * 4)); CK(hipMalloc(&v1, (n + 1) * 4));
    CK(hipMemcpy(k0, keys.data(), n * sizeof(K), hipMemcpyHostToDevice));
    CK(hipMemcpy(v0, vals.data(), n * 4, hipMemcpyHostToDevice));
    size_t bytes = 0;
    CK(smvp::prim::radix_sort_pairs((void *)nullptr, bytes, (const K *)k0, k1, (const unsigned *)v0, v1, n, begin_bit, end_bit, nullptr));
    void *tmp;
    CK(hipMalloc(&tmp, bytes));
    CK(smvp::prim::radix_sort_pairs(tmp, bytes, (const K *)k0, k1, (const unsigned *)v0, v1, n, begin_bit, end_bit, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<K> gk(n), in_after(n);
    std::vector<unsigned> gv(n);
    CK(hipMemcpy(gk.data(), k1, n * sizeof(K), hipMemcpyDeviceToHost));
    CK(hipMemcpy(gv.data(), v1, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(in_after.data(), k0, n * sizeof(K), hipMemcpyDeviceToHost));
    std::vector<unsigned> order(n);
    std::iota(order.begin(), order.end(), 0u);
    const unsigned width = end_bit - begin_bit;
    const unsigned long long m = width >= 64 ? ~0ull : ((1ull << width) - 1);
    auto digit = [&](unsigned i) { return ((unsigned long long)keys[i] >> begin_bit) & m; };
    std::stable_sort(order.begin(), order.end(), [&](unsigned a, unsigned b) { return digit(a) < digit(b); });
    bool ok = in_after == keys;  // the input is not written
    for (size_t i = 0; i < n && ok; ++i)
        ok = gv[i] == order[i] && gk[i] == keys[order[i]];
    if (!ok) {
        ++g_bad;
        printf("SORT MISMATCH: %zu keys of %zu bytes, bits [%u, %u), spread %u\n", n, sizeof(K), begin_bit, end_bit, key_spread_bits);
    }
    CK(hipFree(k0)); CK(hipFree(k1)); CK(hipFree(v0)); CK(hipFree(v1)); CK(hipFree(tmp));
}

static void check_scan(size_t n, bool inclusive, bool in_place, int init, unsigned seed)
{
    std::mt19937 rng(seed);
    std::vector<int> in(n), want(n);
    long long run = init;
    for (size_t i = 0; i < n; ++i) {
        in[i] = (int)(rng() % 7);
        if (inclusive) { run += in[i]; want[i] = (int)run; } else { want[i] = (int)run; run += in[i]; }
    }
    int *d_in, *d_out;
    CK(hipMalloc(&d_in, (n + 1) * 4)); CK(hipMalloc(&d_out, (n + 1) * 4));
    CK(hipMemcpy(d_in, in.data(), n * 4, hipMemcpyHostToDevice));
    int *out = in_place ? d_in : d_out;
    size_t bytes = 0;
    if (inclusive) CK(smvp::prim::inclusive_scan(nullptr, bytes, d_in, out, n, nullptr));
    else CK(smvp::prim::exclusive_scan(nullptr, bytes, d_in, out, init, n, nullptr));
    void *tmp;
    CK(hipMalloc(&tmp, bytes));
    if (inclusive) CK(smvp::prim::inclusive_scan(tmp, bytes, d_in, out, n, nullptr));
    else CK(smvp::prim::exclusive_scan(tmp, bytes, d_in, out, init, n, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<int> got(n);
    CK(hipMemcpy(got.data(), out, n * 4, hipMemcpyDeviceToHost));
    if (got != want) {
        ++g_bad;
        size_t i = 0;
        while (i < n && got[i] == want[i]) ++i;
        printf("SCAN MISMATCH: n %zu inclusive %d in_place %d init %d: first at %zu (%d, want %d)\n", n, inclusive, in_place, init, i, got[i], want[i]);
    }
    CK(hipFree(d_in)); CK(hipFree(d_out)); CK(hipFree(tmp));
}

// `prim_check time`: milliseconds of one sort of 2^27 pairs (64-bit keys, 44 bits) and of one scan of 2^27 ints
static void timing()
{
    const size_t n = (size_t)1 << 27;
    unsigned long long *k0, *k1;
    unsigned *v0, *v1;
    CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
    std::vector<unsigned long long> h(n);
    std::mt19937_64 rng(7);
    for (auto &x : h) x = rng() & ((1ull << 44) - 1);
    CK(hipMemcpy(k0, h.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemset(v0, 0, n * 4));
    size_t bytes = 0;
    CK(smvp::prim::radix_sort_pairs((void *)nullptr, bytes, (const unsigned long long *)k0, k1, (const unsigned *)v0, v1, n, 0u, 44u, nullptr));
    void *tmp;
    CK(hipMalloc(&tmp, bytes));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        CK(smvp::prim::radix_sort_pairs(tmp, bytes, (const unsigned long long *)k0, k1, (const unsigned *)v0, v1, n, 0u, 44u, nullptr));
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("sort 2^27 pairs, 44 bits, %d items per thread: %.2f ms\n", smvp::prim::kSortItems, ms);
    }
    size_t sb = 0;
    CK(smvp::prim::exclusive_scan(nullptr, sb, (const int *)v0, (int *)v1, 0, n, nullptr));
    void *st; CK(hipMalloc(&st, sb));
    CK(hipEventRecord(a));
    CK(smvp::prim::exclusive_scan(st, sb, (const int *)v0, (int *)v1, 0, n, nullptr));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("exclusive scan of 2^27 ints: %.2f ms\n", ms);
}

int main(int argc, char **)
{
    if (argc > 1) {
        timing();
        return 0;
    }
    const size_t sizes[] = {0, 1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 2047, 2048, 2049, 4095, 4097, 100000, 1048576 + 3, 5000000 + 17};
    unsigned seed = 1;
    for (size_t n : sizes) {
        check_sort<unsigned long long>(n, 0, 64, 64, ++seed);
        check_sort<unsigned long long>(n, 0, 41, 41, ++seed);   // the converters' (major, minor) keys
        check_sort<unsigned long long>(n, 0, 13, 3, ++seed);    // few distinct keys: long runs of equal digits (stability)
        check_sort<unsigned long long>(n, 5, 30, 40, ++seed);   // a window of bits: the others must not matter
        check_sort<unsigned>(n, 0, 32, 32, ++seed);
        check_sort<unsigned>(n, 0, 9, 9, ++seed);
        check_sort<unsigned>(n, 0, 0, 20, ++seed);              // no bits: a copy
        for (int mode = 0; mode < 4; ++mode)
            check_scan(n, mode & 1, mode & 2, mode == 0 ? 5 : 0, ++seed);
    }
    check_scan(20000000 + 11, false, true, 0, ++seed);          // three levels of tile totals
    check_scan(1024 * 1024 + 1, true, false, 0, ++seed);
    if (g_bad)
        printf("%d check(s) failed\n", g_bad);
    else
        printf("prim ok\n");
    return g_bad ? 1 : 0;
}
