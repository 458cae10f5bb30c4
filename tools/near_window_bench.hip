// near_window_bench.hip -- stand-alone experiment (not part of the library): the NEAR part of the binned plan (entries within
// 4096 of the diagonal) with a row block's window of x in LDS instead of gathers served by the L2.
//
//   one workgroup (1024 threads) per block of 8192 rows; x[R0 - 4096, R0 + 8192 + 4096) in LDS (128 KB) + the block's row
//   order (16 KB) + its long rows (4 KB);
//   the block's rows sorted by length (plan), 64 sorted rows = one slice, stored step by step (a step = one entry of each
//   of the 64 rows, lane = row); a row longer than `cap` entries is a slice of its own (lane = every 64th entry).  Every
//   wavefront owns a contiguous run of steps -- its short slices, then its long rows -- and walks it as ONE flat stream of
//   8-byte values + 16-bit words (column inside the window | valid | last step of the slice), two batches of 8 steps in
//   flight: every lane sums its own row left to right in a register, no products in LDS, no barrier after the window's.
//
// Built as a shared library and driven by tools/exp_near_window.py (the plan is built there with numpy):
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC tools/near_window_bench.hip -o tools/libnear_window_bench.so
#include <hip/hip_runtime.h>

#include <cstdio>

namespace {

#ifndef NW_RB
#define NW_RB 8192
#endif
#ifndef NW_THREADS
#define NW_THREADS 1024
#endif
#ifndef NW_LONGCAP
#define NW_LONGCAP 1024
#endif
constexpr int kRB = NW_RB, kBand = 4096, kWin = kRB + 2 * kBand, kThreads = NW_THREADS, kWaves = kThreads / 64;
#ifndef NW_U
#define NW_U 8
#endif
constexpr int kU = NW_U, kLongCap = NW_LONGCAP, kPerWave = kRB / 64 / kWaves;
constexpr int kValid = 0x8000, kEnd = 0x4000, kColMask = 0x3fff;

typedef double double2v __attribute__((ext_vector_type(2)));

__device__ inline double wave_sum_fixed(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o);
    return v;
}

// variant: 0 = the product; 1 = timing only, y stored in slice order (coalesced) instead of row order; 2 = the product with
// ordinary instead of non-temporal y stores; 3 = the product, every lane keeps its rows' sums until the block's stream has ended,
// then the window's LDS becomes the block's y and is stored coalesced; 4 / 5 = timing only: 3 without the window's load / without
// the y phase at the block's end
template <int VARIANT>
__global__ __launch_bounds__(kThreads) void near_window(const double *__restrict__ x, double *__restrict__ y, int rows, int cols,
                                                       const long long *__restrict__ wave_ptr, const int *__restrict__ wave_n1,
                                                       const int *__restrict__ wave_n2, const int *__restrict__ blk_short,
                                                       const unsigned short *__restrict__ perm16, const double *__restrict__ sval,
                                                       const unsigned short *__restrict__ sword, const int *__restrict__ blk_long_ptr,
                                                       const unsigned short *__restrict__ long_row16, int xcd, int nblocks_arg)
{
    extern __shared__ double lds[];
    double *xw = lds;
    unsigned short *perm = reinterpret_cast<unsigned short *>(lds + kWin);
    unsigned short *longs = perm + kRB;
    double *lsum = reinterpret_cast<double *>(longs + kLongCap);  // variant 3: the long rows' sums until the window is free
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    int b = blockIdx.x;
    if (xcd) {  // XCD i (workgroups i, i + 8, ...) takes the i-th eighth of the row blocks, in order
        const int per = ((int)gridDim.x + 7) >> 3;
        b = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
        if (b >= nblocks_arg)
            return;
    }
    const long long R0 = (long long)b * kRB;
    const long long wbase = R0 - kBand > 0 ? R0 - kBand : 0;
    const long long wend = R0 + kRB + kBand < (long long)cols ? R0 + kRB + kBand : (long long)cols;
    const int wlen = (int)(wend - wbase);

    const int gw = b * kWaves + wave;
    const long long off = wave_ptr[gw];
    const int n1 = wave_n1[gw], n = n1 + wave_n2[gw];
    const int lq0 = blk_long_ptr[b], nlong = blk_long_ptr[b + 1] - lq0;
    const int nshort = blk_short[b];  // slices of this block that hold at least one entry
    const double *pv = sval + off * 64 + lane;
    const unsigned short *pw = sword + off * 64 + lane;

    double v[2][kU];
    int c[2][kU];
    auto request = [&](int j0, int buf) {
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (j0 + u < n) {
                v[buf][u] = __builtin_nontemporal_load(pv + (size_t)(j0 + u) * 64);
                c[buf][u] = __builtin_nontemporal_load(pw + (size_t)(j0 + u) * 64);
            }
    };
    request(0, 0);
    // the window of x, the block's row order, its long rows
    if constexpr (VARIANT != 4)
    for (int i = 2 * t; i + 1 < wlen; i += 2 * kThreads)
        *reinterpret_cast<double2 *>(&xw[i]) = *reinterpret_cast<const double2 *>(x + wbase + i);
    if ((wlen & 1) && t == 0)
        xw[wlen - 1] = x[wbase + wlen - 1];
    for (int i = 4 * t; i < kRB; i += 4 * kThreads)
        *reinterpret_cast<uint2 *>(&perm[i]) = *reinterpret_cast<const uint2 *>(perm16 + (size_t)b * kRB + i);
    for (int i = t; i < nlong && i < kLongCap; i += kThreads)
        longs[i] = long_row16[lq0 + i];
    __syncthreads();
    // rows of slices without entries
    if constexpr (VARIANT < 3)
        for (int p = nshort * 64 + t; p < kRB; p += kThreads)
            if (perm[p] != 0xffff)
                y[R0 + perm[p]] = 0.0;

    double acc = 0.0;
    double accs[kPerWave];  // variant 3: this lane's finished rows
#pragma unroll
    for (int k = 0; k < kPerWave; ++k)
        accs[k] = 0.0;
    int ks = 0, kl = 0;  // slices / long rows of this wavefront finished so far
    auto batch = [&](int j0, int buf) {
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (j0 + u < n) {
                const int word = c[buf][u];
                if (word & kValid)
                    acc += v[buf][u] * xw[word & kColMask];
                if (__builtin_amdgcn_readfirstlane(word) & kEnd) {
                    if (j0 + u < n1) {
                        const int p = (wave + kWaves * ks) * 64 + lane;
                        const int r = perm[p];
                        if constexpr (VARIANT >= 3) {
#pragma unroll
                            for (int k = 0; k < kPerWave; ++k)
                                if (ks == k)
                                    accs[k] = acc;
                        } else if (r != 0xffff) {
                            if constexpr (VARIANT == 2)
                                y[R0 + r] = acc;
                            else
                                __builtin_nontemporal_store(acc, &y[R0 + (VARIANT == 1 ? p : r)]);
                        }
                        ++ks;
                    } else {
                        const int q = wave + kWaves * kl;
                        const double s = wave_sum_fixed(acc);
                        if constexpr (VARIANT >= 3) {
                            if (lane == 0)
                                lsum[q] = s;  // (the experiment's blocks have at most kLongCap long rows: checked by the driver)
                        } else if (lane == 0)
                            y[R0 + (q < kLongCap ? longs[q] : long_row16[lq0 + q])] = s;
                        ++kl;
                    }
                    acc = 0.0;
                }
            }
    };
    for (int j = 0; j < n; j += 2 * kU) {
        request(j + kU, 1);
        batch(j, 0);
        request(j + 2 * kU, 0);
        batch(j + kU, 1);
    }
    if constexpr (VARIANT == 5) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < kPerWave; ++k)
            s += accs[k];
        if (s == 123.456)
            y[R0 + t] = s;
    }
    if constexpr (VARIANT == 3 || VARIANT == 4) {
        // everybody has finished with the window: it becomes the block's y, filled in row order and stored coalesced
        __syncthreads();
        double *yb = xw;
#pragma unroll
        for (int k = 0; k < kPerWave; ++k) {
            const int r = perm[(wave + kWaves * k) * 64 + lane];
            if (r != 0xffff)
                yb[r] = accs[k];  // (rows of slices without entries: 0)
        }
        for (int q = t; q < nlong; q += kThreads)
            yb[longs[q]] = lsum[q];
        __syncthreads();
        const int nrow = (long long)rows - R0 < kRB ? (int)(rows - R0) : kRB;
        for (int i = 2 * t; i + 1 < nrow; i += 2 * kThreads)
            __builtin_nontemporal_store(*reinterpret_cast<const double2v *>(&yb[i]), reinterpret_cast<double2v *>(y + R0 + i));
    }
}

}  // namespace

// reps launches between two events on the null stream; returns the average ms (reps == 0: one launch, untimed)
extern "C" float near_window_run(const double *x, double *y, int rows, int cols, const long long *wave_ptr, const int *wave_n1,
                                 const int *wave_n2, const int *blk_short, const unsigned short *perm16, const double *sval,
                                 const unsigned short *sword, const int *blk_long_ptr, const unsigned short *long_row16, int nblocks,
                                 int reps, int variant, int xcd)
{
    const size_t lds = sizeof(double) * kWin + 2 * kRB + 2 * kLongCap + 8 * kLongCap;
    hipError_t e = hipFuncSetAttribute((const void *)near_window<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)near_window<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)near_window<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)near_window<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)near_window<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)near_window<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        printf("hipFuncSetAttribute(%zu) failed: %s\n", lds, hipGetErrorString(e));
        return -1.f;
    }
    auto launch = [&] {
        if (variant == 5)
            hipLaunchKernelGGL(near_window<5>, dim3((unsigned)(xcd ? (nblocks + 7) / 8 * 8 : nblocks)), dim3(kThreads), lds, 0, x, y, rows, cols, wave_ptr, wave_n1, wave_n2,
                               blk_short, perm16, sval, sword, blk_long_ptr, long_row16, xcd, nblocks);
        else if (variant == 4)
            hipLaunchKernelGGL(near_window<4>, dim3((unsigned)(xcd ? (nblocks + 7) / 8 * 8 : nblocks)), dim3(kThreads), lds, 0, x, y, rows, cols, wave_ptr, wave_n1, wave_n2,
                               blk_short, perm16, sval, sword, blk_long_ptr, long_row16, xcd, nblocks);
        else if (variant == 3)
            hipLaunchKernelGGL(near_window<3>, dim3((unsigned)(xcd ? (nblocks + 7) / 8 * 8 : nblocks)), dim3(kThreads), lds, 0, x, y, rows, cols, wave_ptr, wave_n1, wave_n2,
                               blk_short, perm16, sval, sword, blk_long_ptr, long_row16, xcd, nblocks);
        else if (variant == 2)
            hipLaunchKernelGGL(near_window<2>, dim3((unsigned)(xcd ? (nblocks + 7) / 8 * 8 : nblocks)), dim3(kThreads), lds, 0, x, y, rows, cols, wave_ptr, wave_n1, wave_n2,
                               blk_short, perm16, sval, sword, blk_long_ptr, long_row16, xcd, nblocks);
        else if (variant == 1)
            hipLaunchKernelGGL(near_window<1>, dim3((unsigned)(xcd ? (nblocks + 7) / 8 * 8 : nblocks)), dim3(kThreads), lds, 0, x, y, rows, cols, wave_ptr, wave_n1, wave_n2,
                               blk_short, perm16, sval, sword, blk_long_ptr, long_row16, xcd, nblocks);
        else
            hipLaunchKernelGGL(near_window<0>, dim3((unsigned)(xcd ? (nblocks + 7) / 8 * 8 : nblocks)), dim3(kThreads), lds, 0, x, y, rows, cols, wave_ptr, wave_n1, wave_n2,
                               blk_short, perm16, sval, sword, blk_long_ptr, long_row16, xcd, nblocks);
    };
    launch();
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) {
        printf("launch failed\n");
        return -1.f;
    }
    if (reps <= 0)
        return 0.f;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        launch();
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms / (float)reps;
}
