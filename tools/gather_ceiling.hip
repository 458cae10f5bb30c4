// gather_bench.hip -- how many random 8-byte gathers per second does an MI355X sustain?
// Development microbenchmark (not part of the library): out[i] = x[idx[i]] with idx uniform
// over a table of 2^k doubles.  Variants: gathers in flight per lane, cache-policy bits, block size.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int ILP, int POLICY>
__global__ __launch_bounds__(256) void gather(const int *__restrict__ idx, const double *__restrict__ x,
                                              double *__restrict__ out, long long n)
{
    const long long stride = (long long)gridDim.x * 256 * ILP;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * ILP; i < n; i += stride) {
        int c[ILP];
        if (ILP == 4) *reinterpret_cast<int4 *>(c) = *reinterpret_cast<const int4 *>(idx + i);
        else if (ILP == 2) *reinterpret_cast<int2 *>(c) = *reinterpret_cast<const int2 *>(idx + i);
        else if (ILP == 8) { *reinterpret_cast<int4 *>(c) = *reinterpret_cast<const int4 *>(idx + i);
                             *reinterpret_cast<int4 *>(c + 4) = *reinterpret_cast<const int4 *>(idx + i + 4); }
        else if (ILP == 16) { for (int q = 0; q < 4; ++q) *reinterpret_cast<int4 *>(c + 4 * q) = *reinterpret_cast<const int4 *>(idx + i + 4 * q); }
        else c[0] = idx[i];
        double v[ILP];
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            if (POLICY == 1) v[k] = __builtin_nontemporal_load(x + c[k]);
            else if (POLICY == 2) v[k] = __hip_atomic_load(x + c[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else v[k] = x[c[k]];
        }
        double s = 0;
#pragma unroll
        for (int k = 0; k < ILP; ++k) s += v[k];
        if (ILP == 1) out[i] = s; else out[i / ILP] = s;
    }
}

template <int ILP, int POLICY>
void run(const char *name, const int *idx, const double *x, double *out, long long n, int blocks)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((gather<ILP, POLICY>), dim3(blocks), dim3(256), 0, 0, idx, x, out, n);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((gather<ILP, POLICY>), dim3(blocks), dim3(256), 0, 0, idx, x, out, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("  %-28s blocks %6d : %7.3f ms  %6.1f G gathers/s\n", name, blocks, ms, n / ms * 1e-6);
}

int main(int argc, char **argv)
{
    const long long n = 1ll << 28;  // gathers per launch
    int *idx; double *x, *out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, n * 8));
    for (int lg : {12, 15, 17, 19}) {
        const long long tab = 1ll << lg;
        CK(hipMalloc(&x, tab * 8)); CK(hipMemset(x, 0, tab * 8));
        std::vector<int> h(n);
        uint64_t s = 88172645463325252ull;
        for (long long i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (uint64_t)tab); }
        CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
        printf("table 2^%d doubles = %lld MB\n", lg, tab * 8 >> 20);
        run<1, 0>("ilp1 plain", idx, x, out, n, 256 * 8);
        run<4, 0>("ilp4 plain", idx, x, out, n, 256 * 8);
        run<8, 0>("ilp8 plain", idx, x, out, n, 256 * 8);
        run<16, 0>("ilp16 plain", idx, x, out, n, 256 * 8);
        run<8, 0>("ilp8 plain", idx, x, out, n, 256 * 16);
        run<8, 0>("ilp8 plain", idx, x, out, n, 256 * 4);
        run<4, 0>("ilp4 plain", idx, x, out, n, 256 * 4);
        run<4, 0>("ilp4 plain", idx, x, out, n, 256 * 32);
        run<4, 1>("ilp4 nontemporal", idx, x, out, n, 256 * 8);
        run<4, 2>("ilp4 sc1 (agent relaxed)", idx, x, out, n, 256 * 8);
        CK(hipFree(x));
    }
    return 0;
}
