// far_binned_bench.hip -- stand-alone experiment (not part of the library): the binned two-pass product of the FAR
// entries of the SURVEY 8(d) random model (46.8 M entries uniform over 2^24 columns: the 39 % of that model's gathers
// that run at the L2-miss rate, profiles/r02_far_gather_split.txt).
//
//   pass A  one workgroup per COLUMN BLOCK: the block of x in LDS, the block's entries streamed as val (8 B) + 16-bit
//           word (local column | first-of-cell flag), every product stored into the bins -- ordered (row block, column
//           block, row, column) -- at stream position + shift[cell]; a cell = (column block, row block).
//   pass B  one workgroup per ROW BLOCK of at most S far entries: its bins are one contiguous run (8 B product + 16-bit
//           slot); the products go to their row-major slot in LDS, one lane per row sums left to right (ascending
//           column: the order of main-cli.c:410-416 among the far entries) and adds to y.
//   Q > 1   pass A's cells are what its speed hangs on (every cell ends in two partly written 128-byte lines), so Q
//           consecutive row blocks form a SUPER block: the bins are ordered (super block, column block, row, column), a
//           cell = (column block, super block) is Q times as long, and a row block of pass B reads, for every column
//           block, its sub-run of the cell (a table of {start, end} per (row block, column block)); the Q row blocks of
//           a super block run together on one XCD so that the lines they share are fetched once.
//
// The plan is built on the host here (the library builds it on the device).  Prints ms per pass for a sweep of column-
// block sizes, row-block sizes and block->XCD orders, and checks the result against the host.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/far_binned_bench.hip -o tools/far_binned_bench.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);      \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

constexpr int kPad = 0xffff;  // 16-bit word of a padding entry of stream A (nothing is stored for it)

// ---------------------------------------------------------------------------------------------------------------------
// pass A.  CBITS: log2 columns per block; THREADS per workgroup; U entries per lane and iteration.
// xcd_group: column blocks are dealt so that XCD i (blocks b with b % 8 == i) works on `xcd_group` neighbouring blocks
// at a time (their cells are neighbours in the bins, so partial lines meet in one L2); 0 = block b takes column block b.
// ---------------------------------------------------------------------------------------------------------------------
template <int CBITS, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void far_pass_a(const double *__restrict__ x, long long ncols,
                                                       const double *__restrict__ a_val, const unsigned short *__restrict__ a_cw,
                                                       const int *__restrict__ chunk_cell, const long long *__restrict__ cb_ptr,
                                                       const int *__restrict__ cell_ptr, const int *__restrict__ cell_shift,
                                                       double *__restrict__ bins, int ncb, int xcd_group, int max_cells)
{
    extern __shared__ double lds[];
    double *xs = lds;
    int *shift = reinterpret_cast<int *>(lds + (1 << CBITS));
    int cb = blockIdx.x;
    if (xcd_group > 0) {
        const int xcd = cb & 7, seq = cb >> 3;
        cb = (seq / xcd_group) * (8 * xcd_group) + xcd * xcd_group + seq % xcd_group;
    }
    if (cb >= ncb)
        return;
    const int t = threadIdx.x;
    const long long a = cb_ptr[cb], z = cb_ptr[cb + 1];  // multiples of 64
    const long long c0 = (long long)cb << CBITS;
    for (int i = t; i < (1 << CBITS); i += THREADS)
        xs[i] = c0 + i < ncols ? x[c0 + i] : 0.0;
    const int cp = cell_ptr[cb], ncell = cell_ptr[cb + 1] - cp;
    for (int i = t; i < ncell && i < max_cells; i += THREADS)
        shift[i] = cell_shift[cp + i];
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    const unsigned long long le = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1);
    constexpr long long STEP = (long long)THREADS * U;
    for (long long base = a + (long long)wave * 64 * U; base < z; base += STEP) {
        double v[U];
        int w[U], cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long j = base + u * 64 + lane;
            const bool in = j < z;
            v[u] = in ? __builtin_nontemporal_load(a_val + j) : 0.0;
            w[u] = in ? (int)__builtin_nontemporal_load(a_cw + j) : kPad;
            cc[u] = in ? chunk_cell[j >> 6] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long j = base + u * 64 + lane;
            const bool real = w[u] != kPad;
            const unsigned long long m = __ballot(real && (w[u] & 0x8000));
            const int cell = cc[u] + __popcll(m & le);
            if (real) {
                const double p = v[u] * xs[w[u] & ((1 << CBITS) - 1)];
                const int sh = cell < max_cells ? shift[cell] : cell_shift[cp + cell];
                __builtin_nontemporal_store(p, bins + (j + sh));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// pass B.  S: far entries a row block holds at most (LDS slots); THREADS per workgroup; U loads per lane in flight.
// ---------------------------------------------------------------------------------------------------------------------
template <int S, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void far_pass_b(const double *__restrict__ bins, const unsigned short *__restrict__ slot,
                                                       const int *__restrict__ rb_row, const int *__restrict__ far_row_ptr,
                                                       double *__restrict__ y, int nrb)
{
    extern __shared__ double fp[];
    const int rb = blockIdx.x;
    if (rb >= nrb)
        return;
    const int t = threadIdx.x;
    const int r0 = rb_row[rb], r1 = rb_row[rb + 1];
    const int a = far_row_ptr[r0], n = far_row_ptr[r1] - a;
    // this lane's first row bounds, requested with the stream
    int ra = 0, rz = 0;
    if (r0 + t < r1) {
        ra = far_row_ptr[r0 + t];
        rz = far_row_ptr[r0 + t + 1];
    }
    for (int base = 0; base < n; base += THREADS * U) {
        double p[U];
        int s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * THREADS + t;
            const bool in = i < n;
            p[u] = in ? __builtin_nontemporal_load(bins + a + i) : 0.0;
            s[u] = in ? (int)__builtin_nontemporal_load(slot + a + i) : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (s[u] >= 0)
                fp[s[u]] = p[u];
    }
    __syncthreads();
    for (int r = r0 + t; r < r1; r += THREADS) {
        const bool pre = r == r0 + t;
        const int sa = (pre ? ra : far_row_ptr[r]) - a, sz = (pre ? rz : far_row_ptr[r + 1]) - a;
        if (sz > sa) {
            double acc = 0.0;
            for (int i = sa; i < sz; ++i)
                acc += fp[i];
            y[r] += acc;
        }
    }
}

// pass B over sub-runs: LPS lanes per (row block, column block) sub-run, U sub-runs per lane group in flight.
template <int S, int THREADS, int LPS, int U>
__global__ __launch_bounds__(THREADS) void far_pass_b_sub(const double *__restrict__ bins, const unsigned short *__restrict__ slot,
                                                           const int2 *__restrict__ subrun, const int *__restrict__ rb_row,
                                                           const int *__restrict__ far_row_ptr, double *__restrict__ y, int nrb,
                                                           int ncb, int q)
{
    extern __shared__ double fp[];
    int rb = blockIdx.x;
    {
        const int xcd = rb & 7, seq = rb >> 3;  // XCD i takes q consecutive row blocks -- one super block -- in a row
        rb = (seq / q) * (8 * q) + xcd * q + seq % q;
    }
    if (rb >= nrb)
        return;
    const int t = threadIdx.x;
    const int r0 = rb_row[rb], r1 = rb_row[rb + 1];
    const int a = far_row_ptr[r0];
    int ra = 0, rz = 0;
    if (r0 + t < r1) {
        ra = far_row_ptr[r0 + t];
        rz = far_row_ptr[r0 + t + 1];
    }
    constexpr int GROUPS = THREADS / LPS;
    const int g = t / LPS, l = t % LPS;
    const int2 *mine = subrun + (size_t)rb * ncb;
    for (int cb0 = 0; cb0 < ncb; cb0 += GROUPS * U) {
        int2 sr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cb = cb0 + u * GROUPS + g;
            sr[u] = cb < ncb ? mine[cb] : make_int2(0, 0);
        }
        double p[U];
        int s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = sr[u].x + l;
            const bool in = j < sr[u].y;
            p[u] = in ? __builtin_nontemporal_load(bins + j) : 0.0;
            s[u] = in ? (int)__builtin_nontemporal_load(slot + j) : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (s[u] >= 0)
                fp[s[u]] = p[u];
#pragma unroll
        for (int u = 0; u < U; ++u)
            for (int j = sr[u].x + l + LPS; j < sr[u].y; j += LPS)
                fp[slot[j]] = bins[j];
    }
    __syncthreads();
    for (int r = r0 + t; r < r1; r += THREADS) {
        const bool pre = r == r0 + t;
        const int sa = (pre ? ra : far_row_ptr[r]) - a, sz = (pre ? rz : far_row_ptr[r + 1]) - a;
        if (sz > sa) {
            double acc = 0.0;
            for (int i = sa; i < sz; ++i)
                acc += fp[i];
            y[r] += acc;
        }
    }
}

// pass B, mirror image of pass A: the row block's entries as ONE virtual stream (its sub-runs one after the other): the
// 16-bit words (slot | first-of-sub-run flag) are stored in that order and read contiguously; entry i's product lies at
// bins[a + i + shift[k]], k = the number of the (non-empty) sub-run it belongs to, counted from the flags like pass A's cells.
template <int S, int THREADS, int U>
__global__ __launch_bounds__(THREADS) void far_pass_b_str(const double *__restrict__ bins, const unsigned short *__restrict__ b_sw,
                                                           const int *__restrict__ b_chunk, const int *__restrict__ b_chunk_ptr,
                                                           const int *__restrict__ b_shift, const int *__restrict__ b_shift_ptr,
                                                           const int *__restrict__ rb_row, const int *__restrict__ far_row_ptr,
                                                           double *__restrict__ y, int nrb, int q, int max_runs)
{
    extern __shared__ double fp[];
    int *shift = reinterpret_cast<int *>(fp + S);
    int rb = blockIdx.x;
    {
        const int xcd = rb & 7, seq = rb >> 3;  // XCD i takes q consecutive row blocks -- one super block -- in a row
        rb = (seq / q) * (8 * q) + xcd * q + seq % q;
    }
    if (rb >= nrb)
        return;
    const int t = threadIdx.x;
    const int r0 = rb_row[rb], r1 = rb_row[rb + 1];
    const int a = far_row_ptr[r0], n = far_row_ptr[r1] - a;
    const int sp = b_shift_ptr[rb], nruns = b_shift_ptr[rb + 1] - sp;
    const int cp = b_chunk_ptr[rb];
    for (int i = t; i < nruns && i < max_runs; i += THREADS)
        shift[i] = b_shift[sp + i];
    int ra = 0, rz = 0;
    if (r0 + t < r1) {
        ra = far_row_ptr[r0 + t];
        rz = far_row_ptr[r0 + t + 1];
    }
    const int lane = t & 63;
    const unsigned long long le = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1);
    __syncthreads();
    for (int base = 0; base < n; base += THREADS * U) {
        int w[U], cc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * THREADS + t;
            const bool in = i < n;
            w[u] = in ? (int)__builtin_nontemporal_load(b_sw + a + i) : -1;
            cc[u] = in ? b_chunk[cp + (i >> 6)] : 0;
        }
        double p[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * THREADS + t;
            const bool in = w[u] >= 0;
            const unsigned long long m = __ballot(in && (w[u] & 0x8000));
            const int k = cc[u] + __popcll(m & le);
            int sh = 0;
            if (in)
                sh = k < max_runs ? shift[k] : b_shift[sp + k];
            p[u] = in ? __builtin_nontemporal_load(bins + (a + i + sh)) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (w[u] >= 0)
                fp[w[u] & 0x3fff] = p[u];
    }
    __syncthreads();
    for (int r = r0 + t; r < r1; r += THREADS) {
        const bool pre = r == r0 + t;
        const int sa = (pre ? ra : far_row_ptr[r]) - a, sz = (pre ? rz : far_row_ptr[r + 1]) - a;
        if (sz > sa) {
            double acc = 0.0;
            for (int i = sa; i < sz; ++i)
                acc += fp[i];
            y[r] += acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
struct Plan {
    int cbits = 14, S = 16384, Q = 1;
    long long nf = 0, na = 0;  // far entries; stream A length (with padding)
    int ncb = 0, nrb = 0, ncells = 0;
    std::vector<double> a_val;
    std::vector<unsigned short> a_cw, slot;
    std::vector<int> chunk_cell, cell_ptr, cell_shift, rb_row;
    std::vector<int2> subrun;
    std::vector<long long> cb_ptr;
    std::vector<unsigned short> b_sw;
    std::vector<int> b_chunk, b_chunk_ptr, b_shift, b_shift_ptr;
};

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static void build_plan(Plan &P, long long nrows, long long ncols, const std::vector<int> &frp, const std::vector<int> &col,
                       const std::vector<double> &val)
{
    const long long nf = frp[nrows];
    P.nf = nf;
    P.ncb = (int)((ncols + (1ll << P.cbits) - 1) >> P.cbits);
    // row blocks: consecutive rows holding at most S far entries
    P.rb_row.clear();
    P.rb_row.push_back(0);
    {
        long long r = 0;
        while (r < nrows) {
            const int a = frp[r];
            long long e = r;
            while (e < nrows && frp[e + 1] - a <= P.S)
                ++e;
            if (e == r) {
                printf("row %lld holds more than S far entries\n", r);
                exit(1);
            }
            P.rb_row.push_back((int)e);
            r = e;
        }
    }
    P.nrb = (int)P.rb_row.size() - 1;
    const int nsb = (P.nrb + P.Q - 1) / P.Q;
    // bins order: inside every SUPER block by (column block, row-major rank); slot = rank inside the (fine) row block;
    // subrun[rb][cb] = where row block rb's entries of column block cb lie in the bins
    std::vector<int> binpos(nf);
    P.slot.assign(nf, 0);
    P.subrun.assign((size_t)P.nrb * P.ncb, make_int2(0, 0));
    {
        std::vector<int> cnt(P.ncb + 1);
        for (int sb = 0; sb < nsb; ++sb) {
            const int rb0 = sb * P.Q, rb1 = std::min(P.nrb, rb0 + P.Q);
            const int a = frp[P.rb_row[rb0]], z = frp[P.rb_row[rb1]];
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int f = a; f < z; ++f)
                ++cnt[(col[f] >> P.cbits) + 1];
            for (int c = 0; c < P.ncb; ++c)
                cnt[c + 1] += cnt[c];
            for (int rb = rb0; rb < rb1; ++rb) {
                const int ra = frp[P.rb_row[rb]], rz = frp[P.rb_row[rb + 1]];
                for (int c = 0; c < P.ncb; ++c)
                    P.subrun[(size_t)rb * P.ncb + c].x = a + cnt[c];
                for (int f = ra; f < rz; ++f) {
                    const int p = a + cnt[col[f] >> P.cbits]++;
                    binpos[f] = p;
                    P.slot[p] = (unsigned short)(f - ra);
                }
                for (int c = 0; c < P.ncb; ++c)
                    P.subrun[(size_t)rb * P.ncb + c].y = a + cnt[c];
            }
        }
    }
    // stream A: by (column block, row-major rank), every column block padded to a multiple of 64
    std::vector<long long> cstart(P.ncb + 1, 0);
    for (long long f = 0; f < nf; ++f)
        ++cstart[(col[f] >> P.cbits) + 1];
    P.cb_ptr.assign(P.ncb + 1, 0);
    for (int c = 0; c < P.ncb; ++c)
        P.cb_ptr[c + 1] = P.cb_ptr[c] + ((cstart[c + 1] + 63) / 64) * 64;
    P.na = P.cb_ptr[P.ncb];
    P.a_val.assign(P.na, 0.0);
    P.a_cw.assign(P.na, (unsigned short)kPad);
    std::vector<int> a_f(P.na, -1);
    {
        std::vector<long long> fill(P.cb_ptr.begin(), P.cb_ptr.end() - 1);
        for (long long f = 0; f < nf; ++f) {
            const int c = col[f] >> P.cbits;
            const long long p = fill[c]++;
            a_f[p] = (int)f;
            P.a_val[p] = val[f];
            P.a_cw[p] = (unsigned short)(col[f] & ((1 << P.cbits) - 1));
        }
    }
    // cells: runs of stream A inside one (column block, super block)
    P.cell_ptr.assign(P.ncb + 1, 0);
    P.cell_shift.clear();
    P.chunk_cell.assign(P.na / 64, 0);
    for (int c = 0; c < P.ncb; ++c) {
        int cells = 0, prev_sb = -1, rb = 0;  // inside a column block the stream ascends in f, so the row block only moves forward
        for (long long p = P.cb_ptr[c]; p < P.cb_ptr[c + 1]; ++p) {
            if ((p & 63) == 0)
                P.chunk_cell[p >> 6] = cells - 1;
            const int f = a_f[p];
            if (f < 0)
                continue;
            while (frp[P.rb_row[rb + 1]] <= f)
                ++rb;
            if (rb / P.Q != prev_sb) {
                P.a_cw[p] |= 0x8000;
                P.cell_shift.push_back((int)(binpos[f] - p));
                ++cells;
                prev_sb = rb / P.Q;
            }
        }
        P.cell_ptr[c + 1] = P.cell_ptr[c] + cells;
    }
    P.ncells = P.cell_ptr[P.ncb];
    // pass B's virtual streams: row block rb's sub-runs (column blocks ascending) one after the other
    P.b_sw.assign(nf, 0);
    P.b_chunk.clear();
    P.b_shift.clear();
    P.b_chunk_ptr.assign(P.nrb + 1, 0);
    P.b_shift_ptr.assign(P.nrb + 1, 0);
    for (int rb = 0; rb < P.nrb; ++rb) {
        const int a = frp[P.rb_row[rb]];
        int i = 0, runs = 0;
        for (int c = 0; c < P.ncb; ++c) {
            const int2 sr = P.subrun[(size_t)rb * P.ncb + c];
            for (int j = sr.x; j < sr.y; ++j, ++i) {
                if ((i & 63) == 0)
                    P.b_chunk.push_back(runs - 1);
                unsigned short w = P.slot[j];
                if (j == sr.x) {
                    w |= 0x8000;
                    P.b_shift.push_back(j - (a + i));
                    ++runs;
                }
                P.b_sw[(size_t)a + i] = w;
            }
        }
        P.b_chunk_ptr[rb + 1] = (int)P.b_chunk.size();
        P.b_shift_ptr[rb + 1] = (int)P.b_shift.size();
    }
}

template <class T>
static T *dev(const std::vector<T> &h)
{
    T *d = nullptr;
    CK(hipMalloc((void **)&d, std::max<size_t>(h.size(), 4) * sizeof(T)));
    CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

struct Dev {
    double *x, *y, *a_val, *bins;
    unsigned short *a_cw, *slot;
    int *chunk_cell, *cell_ptr, *cell_shift, *rb_row, *frp;
    int2 *subrun;
    long long *cb_ptr;
    unsigned short *b_sw;
    int *b_chunk, *b_chunk_ptr, *b_shift, *b_shift_ptr;
};

static float timed(int reps, const std::function<void()> &launch)
{
    if (reps == 0) {  // one launch, for a check
        launch();
        CK(hipDeviceSynchronize());
        return 0.f;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i)
        launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms / reps;
}

template <int CBITS, int THREADS, int U>
static float time_a(const Plan &P, const Dev &D, long long ncols, int xcd_group, int reps)
{
    const int max_cells = 2048;
    const size_t lds = sizeof(double) * (1u << CBITS) + sizeof(int) * max_cells;
    CK(hipFuncSetAttribute((const void *)far_pass_a<CBITS, THREADS, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = xcd_group > 0 ? ((P.ncb + 8 * xcd_group - 1) / (8 * xcd_group)) * 8 * xcd_group : P.ncb;
    return timed(reps, [&] {
        hipLaunchKernelGGL((far_pass_a<CBITS, THREADS, U>), dim3(grid), dim3(THREADS), lds, 0, D.x, ncols, D.a_val, D.a_cw,
                           D.chunk_cell, D.cb_ptr, D.cell_ptr, D.cell_shift, D.bins, P.ncb, xcd_group, max_cells);
    });
}

template <int S, int THREADS, int LPS, int U>
static float time_b(const Plan &P, const Dev &D, int reps)
{
    const size_t lds = sizeof(double) * S;
    CK(hipFuncSetAttribute((const void *)far_pass_b_sub<S, THREADS, LPS, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = ((P.nrb + 8 * P.Q - 1) / (8 * P.Q)) * 8 * P.Q;
    return timed(reps, [&] {
        hipLaunchKernelGGL((far_pass_b_sub<S, THREADS, LPS, U>), dim3(grid), dim3(THREADS), lds, 0, D.bins, D.slot, D.subrun, D.rb_row,
                           D.frp, D.y, P.nrb, P.ncb, P.Q);
    });
}

template <int S, int THREADS, int U>
static float time_b_str(const Plan &P, const Dev &D, int reps)
{
    const int max_runs = 2048;
    const size_t lds = sizeof(double) * S + sizeof(int) * max_runs;
    CK(hipFuncSetAttribute((const void *)far_pass_b_str<S, THREADS, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = ((P.nrb + 8 * P.Q - 1) / (8 * P.Q)) * 8 * P.Q;
    return timed(reps, [&] {
        hipLaunchKernelGGL((far_pass_b_str<S, THREADS, U>), dim3(grid), dim3(THREADS), lds, 0, D.bins, D.b_sw, D.b_chunk, D.b_chunk_ptr,
                           D.b_shift, D.b_shift_ptr, D.rb_row, D.frp, D.y, P.nrb, P.Q, max_runs);
    });
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 24;
    const long long nrows = 1ll << lg, ncols = nrows;
    const double per_row = argc > 2 ? atof(argv[2]) : 2.787;  // 46.76 M far entries over 2^24 rows
    printf("# far part of the SURVEY 8(d) random model as uniform (row, col) pairs: 2^%d rows, %.3f far entries per row\n", lg, per_row);
    // far entries in row-major (CSR) order, columns ascending inside a row
    std::vector<int> frp(nrows + 1, 0);
    for (long long r = 0; r < nrows; ++r) {
        const double u = (double)(rnd() >> 11) * (1.0 / 9007199254740992.0);
        frp[r + 1] = frp[r] + (int)std::floor(per_row + u);
    }
    const long long nf = frp[nrows];
    std::vector<int> col(nf);
    std::vector<double> val(nf), x(ncols);
    for (long long r = 0; r < nrows; ++r) {
        for (int f = frp[r]; f < frp[r + 1]; ++f)
            col[f] = (int)(rnd() % (uint64_t)ncols);
        std::sort(col.begin() + frp[r], col.begin() + frp[r + 1]);
    }
    for (auto &v : val)
        v = (double)(rnd() >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    for (auto &v : x)
        v = (double)(rnd() >> 11) * (1.0 / 9007199254740992.0);
    std::vector<double> y_ref(nrows, 0.0);
    for (long long r = 0; r < nrows; ++r) {
        double acc = 0.0;
        for (int f = frp[r]; f < frp[r + 1]; ++f)
            acc += val[f] * x[col[f]];
        y_ref[r] = acc;
    }
    printf("# %lld far entries\n", nf);

    Dev D;
    D.x = dev(x);
    CK(hipMalloc((void **)&D.y, nrows * sizeof(double)));
    D.frp = dev(frp);
    CK(hipMalloc((void **)&D.bins, (nf + 64) * sizeof(double)));

    struct Cfg { int cbits, S, Q; };
    const Cfg cfgs[] = {{14, 16384, 1}, {14, 16384, 4}, {14, 16384, 8}, {14, 16384, 16}, {14, 8192, 8}, {14, 8192, 16}, {13, 16384, 16}};
    for (const Cfg &c : cfgs) {
        Plan P;
        P.cbits = c.cbits, P.S = c.S, P.Q = c.Q;
        build_plan(P, nrows, ncols, frp, col, val);
        D.a_val = dev(P.a_val), D.a_cw = dev(P.a_cw), D.slot = dev(P.slot), D.chunk_cell = dev(P.chunk_cell);
        D.cell_ptr = dev(P.cell_ptr), D.cell_shift = dev(P.cell_shift), D.rb_row = dev(P.rb_row), D.cb_ptr = dev(P.cb_ptr);
        D.subrun = dev(P.subrun);
        D.b_sw = dev(P.b_sw), D.b_chunk = dev(P.b_chunk), D.b_chunk_ptr = dev(P.b_chunk_ptr), D.b_shift = dev(P.b_shift), D.b_shift_ptr = dev(P.b_shift_ptr);
        printf("columns per block 2^%d (%d blocks), S = %d x Q = %d (%d row blocks): %d cells of %.1f entries, %.1f per sub-run\n",
               c.cbits, P.ncb, c.S, c.Q, P.nrb, P.ncells, (double)nf / P.ncells, (double)nf / ((double)P.nrb * P.ncb));
        const double bytes_a = nf * 18.0 + ncols * 8.0, bytes_b = nf * 10.0 + nrows * 20.0 + 8.0 * P.nrb * P.ncb;
        auto show_a = [&](const char *what, float ms) {
            printf("   pass A %-34s %7.4f ms   %6.0f GB/s of 18 B/entry + x\n", what, ms, bytes_a / ms * 1e-6);
        };
        auto show_b = [&](const char *what, float ms) {
            printf("   pass B %-34s %7.4f ms   %6.0f GB/s of 10 B/entry + 20 B/row + 8 B/sub-run\n", what, ms, bytes_b / ms * 1e-6);
        };
        const int reps = 10;
        if (c.cbits == 14) {
            show_a("1024 thr U8 xcd_group 32", time_a<14, 1024, 8>(P, D, ncols, 32, reps));
            show_a("1024 thr U8 xcd_group 0", time_a<14, 1024, 8>(P, D, ncols, 0, reps));
            show_a("1024 thr U4 xcd_group 32", time_a<14, 1024, 4>(P, D, ncols, 32, reps));
            show_a("1024 thr U16 xcd_group 32", time_a<14, 1024, 16>(P, D, ncols, 32, reps));
        } else {
            show_a("512 thr U8 xcd_group 32", time_a<13, 512, 8>(P, D, ncols, 32, reps));
            show_a("512 thr U8 xcd_group 0", time_a<13, 512, 8>(P, D, ncols, 0, reps));
            show_a("1024 thr U4 xcd_group 32", time_a<13, 1024, 4>(P, D, ncols, 32, reps));
            show_a("512 thr U16 xcd_group 32", time_a<13, 512, 16>(P, D, ncols, 32, reps));
        }
        // one clean product for the check: bins poisoned, y = 0, A, B
        CK(hipMemset(D.bins, 0xff, nf * sizeof(double)));
        if (c.cbits == 14)
            (void)time_a<14, 1024, 8>(P, D, ncols, 32, 1);
        else
            (void)time_a<13, 512, 8>(P, D, ncols, 32, 1);
        auto check = [&](const char *what, const std::function<void()> &launch) {
            CK(hipMemset(D.y, 0, nrows * sizeof(double)));
            launch();
            CK(hipDeviceSynchronize());
            std::vector<double> y(nrows);
            CK(hipMemcpy(y.data(), D.y, nrows * sizeof(double), hipMemcpyDeviceToHost));
            long long bad = 0;
            for (long long r = 0; r < nrows; ++r)
                bad += y[r] != y_ref[r];
            printf("   check %s: %lld of %lld rows differ from the host's left-to-right sums\n", what, bad, nrows);
        };
        const int gridb = ((P.nrb + 8 * P.Q - 1) / (8 * P.Q)) * 8 * P.Q;
        if (c.S == 16384) {
            check("S 16384", [&] {
                CK(hipFuncSetAttribute((const void *)far_pass_b_sub<16384, 1024, 16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
                hipLaunchKernelGGL((far_pass_b_sub<16384, 1024, 16, 4>), dim3(gridb), dim3(1024), 16384 * 8, 0, D.bins, D.slot, D.subrun,
                                   D.rb_row, D.frp, D.y, P.nrb, P.ncb, P.Q);
            });
            show_b("1024 thr, 16 lanes x 4", time_b<16384, 1024, 16, 4>(P, D, reps));
            show_b("1024 thr, 16 lanes x 8", time_b<16384, 1024, 16, 8>(P, D, reps));
            show_b("1024 thr, 32 lanes x 4", time_b<16384, 1024, 32, 4>(P, D, reps));
            show_b("1024 thr, 8 lanes x 4", time_b<16384, 1024, 8, 4>(P, D, reps));
            check("S 16384 stream form", [&] { (void)time_b_str<16384, 1024, 8>(P, D, 0); });
            show_b("stream form 1024 thr U4", time_b_str<16384, 1024, 4>(P, D, reps));
            show_b("stream form 1024 thr U8", time_b_str<16384, 1024, 8>(P, D, reps));
            show_b("stream form 1024 thr U16", time_b_str<16384, 1024, 16>(P, D, reps));
            show_b("stream form 512 thr U16", time_b_str<16384, 512, 16>(P, D, reps));
        } else {
            check("S 8192", [&] {
                CK(hipFuncSetAttribute((const void *)far_pass_b_sub<8192, 512, 8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8));
                hipLaunchKernelGGL((far_pass_b_sub<8192, 512, 8, 4>), dim3(gridb), dim3(512), 8192 * 8, 0, D.bins, D.slot, D.subrun, D.rb_row,
                                   D.frp, D.y, P.nrb, P.ncb, P.Q);
            });
            show_b("512 thr, 8 lanes x 4", time_b<8192, 512, 8, 4>(P, D, reps));
            show_b("512 thr, 8 lanes x 8", time_b<8192, 512, 8, 8>(P, D, reps));
            show_b("512 thr, 16 lanes x 4", time_b<8192, 512, 16, 4>(P, D, reps));
            show_b("1024 thr, 8 lanes x 4", time_b<8192, 1024, 8, 4>(P, D, reps));
            show_b("1024 thr, 16 lanes x 4", time_b<8192, 1024, 16, 4>(P, D, reps));
            check("S 8192 stream form", [&] { (void)time_b_str<8192, 512, 8>(P, D, 0); });
            show_b("stream form 512 thr U4", time_b_str<8192, 512, 4>(P, D, reps));
            show_b("stream form 512 thr U8", time_b_str<8192, 512, 8>(P, D, reps));
            show_b("stream form 512 thr U16", time_b_str<8192, 512, 16>(P, D, reps));
            show_b("stream form 1024 thr U8", time_b_str<8192, 1024, 8>(P, D, reps));
        }
        for (void *p : {(void *)D.a_val, (void *)D.a_cw, (void *)D.slot, (void *)D.chunk_cell, (void *)D.cell_ptr, (void *)D.cell_shift,
                        (void *)D.rb_row, (void *)D.cb_ptr, (void *)D.subrun, (void *)D.b_sw, (void *)D.b_chunk, (void *)D.b_chunk_ptr,
                        (void *)D.b_shift, (void *)D.b_shift_ptr})
            CK(hipFree(p));
        fflush(stdout);
    }
    return 0;
}
