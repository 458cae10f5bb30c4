// gather_spacing.hip -- L2-resident 8-byte gathers: how does the rate depend on how far apart the 64 addresses of one
// wave instruction lie?  Development microbenchmark (not part of the library).  The 64 lanes of an instruction gather
// ascending addresses with a mean gap of g doubles (uniform in [1, 2g-1]) inside a 2 MB table (mod its size): g = 153 is
// what a 2048-row strip of BASELINE config 4 sees in the column sweep, g = 19 what one stream over a CU's 16 K rows would.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_spacing.hip -o tools/gather_spacing.bin && tools/gather_spacing.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void gather4(const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out, long long n)
{
    const long long stride = (long long)gridDim.x * 256 * 4;
    double s = 0;
    for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n; i += stride) {
        int c[4];
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = idx[i + k * 256];      // lane-consecutive entries per instruction
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = x[c[k]];
        s += (v[0] + v[1]) + (v[2] + v[3]);
    }
    out[(long long)blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const long long n = 1ll << 27;
    const long long tab = 1ll << 18;   // 2 MB: L2-resident in every XCD
    int *idx; double *x, *out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, 2048 * 256 * 8)); CK(hipMalloc(&x, tab * 8)); CK(hipMemset(x, 0, tab * 8));
    std::vector<int> h(n);
    for (int g : {1, 2, 4, 8, 19, 38, 76, 153, 600, 4000}) {
        uint64_t s = 88172645463325252ull;
        long long pos = 0;
        for (long long i = 0; i < n; ++i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            if ((i & 63) == 0) pos = (long long)(s % (uint64_t)tab);       // every instruction starts somewhere else
            else pos += 1 + (long long)(s % (uint64_t)(2 * g - 1 > 0 ? 2 * g - 1 : 1));
            h[i] = (int)(pos % tab);
        }
        CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(gather4, dim3(2048), dim3(256), 0, 0, idx, x, out, n);
        CK(hipEventRecord(a));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(gather4, dim3(2048), dim3(256), 0, 0, idx, x, out, n);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        printf("mean gap %5d doubles: %7.3f ms  %6.1f G gathers/s\n", g, ms, n / ms * 1e-6);
    }
    return 0;
}
