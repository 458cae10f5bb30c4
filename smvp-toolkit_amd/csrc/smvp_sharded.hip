// smvp_sharded.hip -- one product over several GPUs of a node from ONE host process.
//
// The reference is a single thread on one CPU; this is new design:
// the matrix is cut into row blocks balanced by entries (smvp_partition_rows), GPU g
// holds block g, all of x, and produces its slice of y; RCCL all-gathers over xGMI
// assemble the full y on every GPU.  Each block is further cut into `chunks` row
// chunks, each its own CSR / TJDS handle: the all-gather of chunk c is put on the GPU's
// communication stream right behind an event of chunk c's product, so it travels while
// chunk c+1 is being multiplied on the compute stream.  With more than one GPU every GPU
// has its own issuing thread (one communicator rank per thread, no grouped calls): the
// launches of eight GPUs are not queued behind each other on one host thread.
// ncclAllGather wants equal counts, so a chunk goes onto the wire padded to
// the tallest chunk c of any GPU and one small kernel per GPU then copies the
// gathered pieces to their rows of the full vector.
// bench.py does the same with one process per GPU through torch.distributed; this
// file is what the C command line (--gpus N) and smvp_*_compute(opts.ngpus > 1) use.
//
// librccl is loaded with dlopen the first time a sharded handle is created, so
// single-GPU runs neither link nor load it.
//
// How the y blocks travel is a measured choice (round 5; SURVEY 7: a one-link ring costs 0.46 ms against 0.1 ms of product,
// direct pushes over all seven links 65 us): SMVP_EXCHANGE_RCCL is ncclAllGather into a padded wire buffer + a placement
// kernel; SMVP_EXCHANGE_COPIES pushes every chunk straight into every rank's full vector with hipMemcpyAsync (the copy
// engines); SMVP_EXCHANGE_DIRECT does the same pushes with one kernel per chunk (stores over xGMI, all links at once);
// SMVP_EXCHANGE_AUTO (the default) times one product's exchange under every form that is available when the handle is
// created and keeps the fastest.  The two push forms order the ranks with events and a host-side meeting point of the
// issuing threads, need no wire buffer and no placement, and accept any device list -- also ONE device several times --
// so that the N-GPU code (issuing threads, chunks, both gather modes, power iteration, early destroy) runs on a one-GPU
// box: "virtual ranks".
#include "smvp_common.h"
#include "smvp_kernels.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return smvp::fail(SMVP_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;  // optional
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    int status = SMVP_ERR_UNSUPPORTED;
    std::string why;
};

int load_rccl(Rccl **out)
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib)
                break;
            const char *err = dlerror();  // one call: it clears the message
            r.why = err ? err : "dlopen failed";
        }
        if (!r.lib)
            return;
        const char *missing = nullptr;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.lib, name);
            if (!p && !missing)
                missing = name;
            return p;
        };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.CommAbort = (decltype(r.CommAbort))dlsym(r.lib, "ncclCommAbort");
        r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
        if (missing)
            r.why = std::string("missing symbol ") + missing;
        else
            r.status = SMVP_OK;
    });
    if (r.status != SMVP_OK)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded: %s", r.why.c_str());
    *out = &r;
    return SMVP_OK;
}

// Runs what follows on `device` and puts the caller's device back afterwards.
struct DeviceScope {
    int prev = -1;
    DeviceScope() { (void)hipGetDevice(&prev); }
    ~DeviceScope()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};

// gathered pieces -> their rows of the full vector.  seg[i] = {first row in y_full, rows, offset in wire, 0}
__global__ __launch_bounds__(256) void place_gathered(const double *__restrict__ wire, double *__restrict__ y_full,
                                                      const int4 *__restrict__ seg, int nseg, int rows)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= rows)
        return;
    int lo = 0, hi = nseg - 1;  // last segment that starts at or before row r (empty segments share a start: take the last)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (seg[mid].x <= r)
            lo = mid;
        else
            hi = mid - 1;
    }
    y_full[r] = wire[(size_t)seg[lo].z + (size_t)(r - seg[lo].x)];
}

// SMVP_EXCHANGE_DIRECT: one rank's chunk into every rank's full vector.  blockIdx.y = destination rank: every destination is
// its own stream of coalesced stores (over that peer's xGMI link; the rank's own copy at HBM speed), the source is re-read
// from L2.  16-byte stores where source and destination are aligned alike, 8-byte otherwise.
struct PushTargets {
    double *y_full[64];
};
__global__ __launch_bounds__(256) void push_chunk(const double *__restrict__ src, const PushTargets dst, long long first_row, int count)
{
    double *__restrict__ out = dst.y_full[blockIdx.y] + first_row;
    const int stride = (int)gridDim.x * 256;
    const int i0 = (int)blockIdx.x * 256 + (int)threadIdx.x;
    const int head = (int)(((uintptr_t)src >> 3) & 1u);  // elements in front of the first 16-byte boundary of the source
    if ((((uintptr_t)out >> 3) & 1u) == (unsigned)head) {
        if (head && i0 == 0 && count > 0)
            out[0] = src[0];
        const int pairs = (count - head) / 2;
        const double2 *s2 = reinterpret_cast<const double2 *>(src + head);
        double2 *o2 = reinterpret_cast<double2 *>(out + head);
        for (int i = i0; i < pairs; i += stride)
            o2[i] = s2[i];
        if (i0 == 0 && head + 2 * pairs < count)
            out[count - 1] = src[count - 1];
    } else {
        for (int i = i0; i < count; i += stride)
            out[i] = src[i];
    }
}

}  // namespace

struct smvp_sharded {
    int format = 0;  // 0 = CSR, 1 = TJDS
    int n = 0, chunks = 1;
    int rows = 0, cols = 0, nnz = 0;
    std::vector<int> device;
    std::vector<int> bounds;      // n + 1 block bounds (rows), balanced by entries
    std::vector<int> cbounds;     // n * (chunks + 1): chunk bounds of every block, global rows
    std::vector<int> pad;         // chunks: tallest chunk c over the GPUs = what travels per GPU for chunk c
    std::vector<size_t> loff;     // chunks + 1: offset of chunk c in a GPU's y_local (padded slots)
    std::vector<size_t> woff;     // chunks + 1: offset of chunk c's n * pad[c] block in the wire buffer
    std::vector<smvp_csr_t *> csr;    // n * chunks
    std::vector<smvp_tjds_t *> tjds;  // n * chunks
    std::vector<hipStream_t> stream, comm_stream;
    std::vector<double *> d_x, d_y_local, d_wire, d_y_full;
    std::vector<int4 *> d_seg;
    int nseg = 0;
    std::vector<hipEvent_t> ev0, ev1, ev_gathered;
    std::vector<hipEvent_t> ev_chunk;  // n * chunks
    std::vector<ncclComm_t> comm;
    std::vector<unsigned long long *> d_norm;
    Rccl *rccl = nullptr;
    // exchange by copies (virtual ranks): ev_pushed[g] = rank g's pieces of this product have landed everywhere,
    // ev_placed[g] = rank g has read its wire buffer (it may be written again); `phase` lines the issuing threads up
    // between recording those events and waiting for them
    int exchange = SMVP_EXCHANGE_RCCL;  // the form in use: never AUTO
    bool auto_exchange = false;         // created with AUTO: a probe keeps the fastest form
    bool have_rccl = false, have_peer = false;  // what can be selected
    double probe_ms[3] = {-1.0, -1.0, -1.0};    // last probe: one product's exchange by form
    bool form_off[3] = {false, false, false};   // a form that failed in a probe is not offered again
    std::string rccl_why;                       // why RCCL is not available (AUTO)
    PushTargets targets;                        // DIRECT: every rank's full vector
    std::vector<hipEvent_t> ev_pushed, ev_placed;
    std::mutex phase_mu;
    std::condition_variable phase_cv;
    int phase_waiting = 0;
    unsigned long long phase_round = 0;
    bool phase_abort = false;  // a rank failed: nobody waits at the meeting point any more
    bool broken = false;  // a rank failed inside a product: its peers' collectives may never complete
    std::string broken_why;

    // n > 1: one issuing thread per GPU, woken for every product (issue_product)
    std::vector<std::thread> issuer;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    unsigned long long job = 0;
    int job_allgather = 0, job_timed = 0, job_skip_products = 0, pending = 0;
    bool quit = false;
    std::vector<int> job_rc;
    std::vector<std::string> job_err;
};

namespace {
void stop_issuers(smvp_sharded *h)
{
    {
        std::lock_guard<std::mutex> lock(h->mu);
        h->quit = true;
    }
    h->cv_go.notify_all();
    for (std::thread &t : h->issuer)
        if (t.joinable())
            t.join();
    h->issuer.clear();
}
}  // namespace

extern "C" void smvp_sharded_destroy(smvp_sharded_t *h)
{
    if (!h)
        return;
    stop_issuers(h);
    DeviceScope keep;
    for (int g = 0; g < (int)h->device.size(); ++g) {
        const size_t i = (size_t)g;
        (void)hipSetDevice(h->device[i]);
        if (i < h->comm.size() && h->comm[i] && h->rccl) {
            if (h->broken && h->rccl->CommAbort)
                h->rccl->CommAbort(h->comm[i]);  // a collective some rank never joined would not let CommDestroy return
            else
                h->rccl->CommDestroy(h->comm[i]);
        }
        for (int c = 0; c < h->chunks; ++c) {
            const size_t k = i * (size_t)h->chunks + (size_t)c;
            if (k < h->csr.size())
                smvp_csr_destroy(h->csr[k]);
            if (k < h->tjds.size())
                smvp_tjds_destroy(h->tjds[k]);
            if (k < h->ev_chunk.size() && h->ev_chunk[k])
                (void)hipEventDestroy(h->ev_chunk[k]);
        }
        for (auto *vec : {&h->d_x, &h->d_y_local, &h->d_wire, &h->d_y_full})
            if (i < vec->size() && (*vec)[i])
                (void)hipFree((*vec)[i]);
        if (i < h->d_seg.size() && h->d_seg[i])
            (void)hipFree(h->d_seg[i]);
        if (i < h->d_norm.size() && h->d_norm[i])
            (void)hipFree(h->d_norm[i]);
        for (auto *vec : {&h->ev0, &h->ev1, &h->ev_gathered, &h->ev_pushed, &h->ev_placed})
            if (i < vec->size() && (*vec)[i])
                (void)hipEventDestroy((*vec)[i]);
        for (auto *vec : {&h->stream, &h->comm_stream})
            if (i < vec->size() && (*vec)[i])
                (void)hipStreamDestroy((*vec)[i]);
    }
    delete h;
}

extern "C" void smvp_shard_opts_default(smvp_shard_opts_t *o)
{
    if (!o)
        return;
    o->struct_size = (unsigned)sizeof *o;
    o->chunks = 0;
    o->balance = 1;
    o->exchange = SMVP_EXCHANGE_AUTO;
}

namespace {

// Block and chunk bounds, per-GPU buffers, streams, events, the communicator.  row_ptr: host CSR row pointer of the
// whole matrix (what the partition is balanced on).
int sharded_common(smvp_sharded *h, int ngpus, const int *devices, int rows, int cols, int nnz, const int *row_ptr,
                   const smvp_shard_opts_t *opts)
{
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
        return smvp::fail(SMVP_ERR_NO_DEVICE, "no HIP device is visible (this engine has no CPU path)");
    smvp_shard_opts_t def;
    smvp_shard_opts_default(&def);
    const smvp_shard_opts_t *o = opts ? opts : &def;
    if (o->struct_size != (unsigned)sizeof(smvp_shard_opts_t))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_shard_opts_t of %u bytes, this library's has %u: initialise it with "
                                            "smvp_shard_opts_default and build against this library's header",
                          o->struct_size, (unsigned)sizeof(smvp_shard_opts_t));
    if (o->exchange < SMVP_EXCHANGE_RCCL || o->exchange > SMVP_EXCHANGE_AUTO)
        return smvp::fail(SMVP_ERR_INVALID, "unknown exchange %d", o->exchange);
    // the push forms, asked for by name, let ranks share a device ("virtual ranks")
    const bool copies = o->exchange == SMVP_EXCHANGE_COPIES || o->exchange == SMVP_EXCHANGE_DIRECT;
    if (ngpus < 1 || (!copies && ngpus > visible) || ngpus > 64)
        return smvp::fail(SMVP_ERR_INVALID, "%d GPUs requested, %d visible", ngpus, visible);
    if (o->chunks < 0 || o->chunks > 64)
        return smvp::fail(SMVP_ERR_INVALID, "chunks per GPU must lie in [1, 64] (0 = default)");
    h->auto_exchange = o->exchange == SMVP_EXCHANGE_AUTO;
    h->exchange = h->auto_exchange ? SMVP_EXCHANGE_RCCL : o->exchange;
    h->n = ngpus;
    h->chunks = o->chunks > 0 ? o->chunks : (ngpus > 1 ? 4 : 1);
    h->rows = rows, h->cols = cols, h->nnz = nnz;
    const size_t n = (size_t)ngpus, C = (size_t)h->chunks;
    for (int g = 0; g < ngpus; ++g) {
        const int dev = devices ? devices[g] : (copies ? g % visible : g);
        if (dev < 0 || dev >= visible)
            return smvp::fail(SMVP_ERR_INVALID, "device %d out of range", dev);
        for (int p : h->device)
            if (p == dev && !copies)
                return smvp::fail(SMVP_ERR_INVALID, "device %d listed twice", dev);
        h->device.push_back(dev);
    }
    // blocks balanced by entries (or of equal height), every block cut into chunks the same way
    h->bounds.assign(n + 1, 0);
    if (o->balance) {
        if (int rc = smvp_partition_rows(row_ptr, rows, ngpus, h->bounds.data()))
            return rc;
    } else {
        for (int g = 0; g <= ngpus; ++g)
            h->bounds[(size_t)g] = (int)((long long)rows * g / ngpus);
    }
    h->cbounds.assign(n * (C + 1), 0);
    h->pad.assign(C, 0);
    for (size_t g = 0; g < n; ++g) {
        const int a = h->bounds[g], b = h->bounds[g + 1];
        std::vector<int> local((size_t)(b - a) + 1), cb(C + 1);
        for (int r = a; r <= b; ++r)
            local[(size_t)(r - a)] = row_ptr[r] - row_ptr[a];
        if (o->balance) {
            if (int rc = smvp_partition_rows(local.data(), b - a, h->chunks, cb.data()))
                return rc;
        } else {
            for (size_t c = 0; c <= C; ++c)
                cb[c] = (int)((long long)(b - a) * (long long)c / (long long)C);
        }
        for (size_t c = 0; c <= C; ++c)
            h->cbounds[g * (C + 1) + c] = a + cb[c];
        for (size_t c = 0; c < C; ++c)
            h->pad[c] = std::max(h->pad[c], cb[c + 1] - cb[c]);
    }
    h->loff.assign(C + 1, 0);
    h->woff.assign(C + 1, 0);
    for (size_t c = 0; c < C; ++c) {
        h->pad[c] = std::max(h->pad[c], 1);
        h->loff[c + 1] = h->loff[c] + (size_t)h->pad[c];
        h->woff[c + 1] = h->woff[c] + (size_t)h->pad[c] * n;
    }
    // where every gathered piece belongs: segments in ascending row order (block-major, chunk-minor)
    std::vector<int4> seg;
    for (size_t g = 0; g < n; ++g)
        for (size_t c = 0; c < C; ++c) {
            const int a = h->cbounds[g * (C + 1) + c], b = h->cbounds[g * (C + 1) + c + 1];
            seg.push_back(make_int4(a, b - a, (int)(h->woff[c] + g * (size_t)h->pad[c]), 0));
        }
    if (h->woff[C] > 2147483647ull)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "the padded wire buffer exceeds 2^31 entries");
    h->nseg = (int)seg.size();

    h->stream.assign(n, nullptr);
    h->comm_stream.assign(n, nullptr);
    for (auto *vec : {&h->d_x, &h->d_y_local, &h->d_wire, &h->d_y_full})
        vec->assign(n, nullptr);
    h->d_seg.assign(n, nullptr);
    for (auto *vec : {&h->ev0, &h->ev1, &h->ev_gathered, &h->ev_pushed, &h->ev_placed})
        vec->assign(n, nullptr);
    h->ev_chunk.assign(n * C, nullptr);
    h->d_norm.assign(n, nullptr);
    const size_t vec_len = (size_t)std::max(std::max(cols, rows), 1);
    for (size_t g = 0; g < n; ++g) {
        HIP_TRY(hipSetDevice(h->device[g]));
        HIP_TRY(hipStreamCreate(&h->stream[g]));
        HIP_TRY(hipStreamCreate(&h->comm_stream[g]));
        for (auto *vec : {&h->ev0, &h->ev1})
            HIP_TRY(hipEventCreate(&(*vec)[g]));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_gathered[g], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_pushed[g], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_placed[g], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(h->ev_placed[g], h->stream[g]));  // (nothing to wait for before the first product)
        for (size_t c = 0; c < C; ++c)
            HIP_TRY(hipEventCreateWithFlags(&h->ev_chunk[g * C + c], hipEventDisableTiming));
        HIP_TRY(hipMalloc((void **)&h->d_x[g], sizeof(double) * vec_len));
        HIP_TRY(hipMalloc((void **)&h->d_norm[g], sizeof(unsigned long long)));
        HIP_TRY(hipMalloc((void **)&h->d_y_local[g], sizeof(double) * h->loff[C]));
        if (!copies)  // the padded wire buffer is RCCL's (the push forms write straight into the full vectors)
            HIP_TRY(hipMalloc((void **)&h->d_wire[g], sizeof(double) * h->woff[C]));
        HIP_TRY(hipMalloc((void **)&h->d_y_full[g], sizeof(double) * vec_len));
        HIP_TRY(hipMalloc((void **)&h->d_seg[g], sizeof(int4) * seg.size()));
        HIP_TRY(hipMemset(h->d_y_local[g], 0, sizeof(double) * h->loff[C]));
        if (!copies)
            HIP_TRY(hipMemset(h->d_wire[g], 0, sizeof(double) * h->woff[C]));
        HIP_TRY(hipMemset(h->d_y_full[g], 0, sizeof(double) * vec_len));
        HIP_TRY(hipMemcpy(h->d_seg[g], seg.data(), sizeof(int4) * seg.size(), hipMemcpyHostToDevice));
    }
    for (size_t g = 0; g < n; ++g)
        h->targets.y_full[g] = h->d_y_full[g];
    bool distinct = true;
    for (size_t g = 0; g < n; ++g)
        for (size_t r = 0; r < g; ++r)
            distinct = distinct && h->device[g] != h->device[r];
    const int want = o->exchange;
    // the push forms: every device must reach its peers' vectors where they differ
    if (want != SMVP_EXCHANGE_RCCL) {
        h->have_peer = true;
        for (size_t g = 0; g < n && h->have_peer; ++g)
            for (size_t r = 0; r < n && h->have_peer; ++r)
                if (h->device[g] != h->device[r]) {
                    HIP_TRY(hipSetDevice(h->device[g]));
                    const hipError_t e = hipDeviceEnablePeerAccess(h->device[r], 0);
                    (void)hipGetLastError();
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                        if (want != SMVP_EXCHANGE_AUTO)
                            return smvp::fail(SMVP_ERR_HIP, "device %d cannot reach device %d: %s", h->device[g], h->device[r],
                                              hipGetErrorString(e));
                        h->have_peer = false;
                    }
                }
    }
    // RCCL: one communicator rank per (distinct) device
    if (want == SMVP_EXCHANGE_RCCL || (want == SMVP_EXCHANGE_AUTO && distinct)) {
        int rc = load_rccl(&h->rccl);
        if (rc == SMVP_OK) {
            h->comm.assign(n, nullptr);
            const ncclResult_t nr = h->rccl->CommInitAll(h->comm.data(), ngpus, h->device.data());
            if (nr != ncclSuccess) {
                rc = smvp::fail(SMVP_ERR_HIP, "ncclCommInitAll failed: %s", h->rccl->GetErrorString(nr));
                h->comm.clear();
            }
        }
        if (rc != SMVP_OK) {
            if (want == SMVP_EXCHANGE_RCCL || !h->have_peer)
                return rc;
            h->rccl_why = smvp_last_error();  // AUTO goes on with the pushes
        } else {
            h->have_rccl = true;
        }
    } else if (want == SMVP_EXCHANGE_AUTO) {
        h->rccl_why = "several ranks share a device";
    }
    if (!h->have_rccl && !h->have_peer)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "no exchange is available (RCCL: %s; no peer access)", h->rccl_why.c_str());
    if (h->auto_exchange)
        h->exchange = h->have_rccl ? SMVP_EXCHANGE_RCCL : SMVP_EXCHANGE_DIRECT;  // until the probe has spoken
    return SMVP_OK;
}

}  // namespace

extern "C" int smvp_csr_sharded_create_ex(smvp_sharded_t **out, int ngpus, const int *devices, int rows, int cols, int nnz,
                                          const int *row_ptr, const int *col_ind, const double *val,
                                          const smvp_shard_opts_t *opts)
{
    if (!out || rows < 0 || cols < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!col_ind || !val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_sharded_create: bad argument");
    if (row_ptr[0] != 0 || row_ptr[rows] != nnz)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_sharded_create: row_ptr does not run from 0 to nnz");
    DeviceScope keep;
    smvp_sharded *h = new smvp_sharded;
    h->format = 0;
    int rc = sharded_common(h, ngpus, devices, rows, cols, nnz, row_ptr, opts);
    const size_t C = (size_t)h->chunks;
    h->csr.assign((size_t)std::max(ngpus, 0) * C, nullptr);
    for (size_t g = 0; rc == SMVP_OK && g < (size_t)ngpus; ++g)
        for (size_t c = 0; rc == SMVP_OK && c < C; ++c) {
            const int a = h->cbounds[g * (C + 1) + c], b = h->cbounds[g * (C + 1) + c + 1];
            std::vector<int> rp((size_t)(b - a) + 1);
            for (int r = a; r <= b; ++r)
                rp[(size_t)(r - a)] = row_ptr[r] - row_ptr[a];
            rc = smvp_csr_create_block(&h->csr[g * C + c], h->device[g], b - a, cols, row_ptr[b] - row_ptr[a], rp.data(),
                                       col_ind + row_ptr[a], val + row_ptr[a], SMVP_MEM_HOST, nullptr, a);
        }
    if (rc == SMVP_OK && h->auto_exchange)
        rc = smvp_sharded_probe_exchange(h, 3);
    if (rc != SMVP_OK) {
        smvp_sharded_destroy(h);
        return rc;
    }
    *out = h;
    return SMVP_OK;
}

extern "C" int smvp_csr_sharded_create(smvp_sharded_t **out, int ngpus, const int *devices, int rows, int cols, int nnz,
                                       const int *row_ptr, const int *col_ind, const double *val)
{
    return smvp_csr_sharded_create_ex(out, ngpus, devices, rows, cols, nnz, row_ptr, col_ind, val, nullptr);
}

extern "C" int smvp_tjds_sharded_create_ex(smvp_sharded_t **out, int ngpus, const int *devices, const smvp_coo_t *coo,
                                           int rows, int cols, int nnz, const smvp_shard_opts_t *opts)
{
    if (!out || rows < 0 || cols < 0 || nnz < 0 || (nnz > 0 && !coo))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_sharded_create: bad argument");
    // entries per row -> the row pointer the partition is balanced on; then every entry goes to its chunk once
    std::vector<int> row_ptr((size_t)rows + 1, 0);
    for (int i = 0; i < nnz; ++i) {
        if (coo[i].row < 0 || coo[i].row >= rows)
            return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_sharded_create: entry %d lies outside the matrix", i);
        ++row_ptr[(size_t)coo[i].row + 1];
    }
    for (int r = 0; r < rows; ++r)
        row_ptr[(size_t)r + 1] += row_ptr[(size_t)r];
    DeviceScope keep;
    smvp_sharded *h = new smvp_sharded;
    h->format = 1;
    int rc = sharded_common(h, ngpus, devices, rows, cols, nnz, row_ptr.data(), opts);
    const size_t C = (size_t)h->chunks, parts = (size_t)std::max(ngpus, 0) * C;
    h->tjds.assign(parts, nullptr);
    if (rc == SMVP_OK) {
        // chunk of every row, then one counting pass buckets the entries (input order kept inside a chunk)
        std::vector<int> chunk_of((size_t)rows), first((size_t)rows);
        std::vector<size_t> start(parts + 1, 0);
        for (size_t g = 0; g < (size_t)ngpus; ++g)
            for (size_t c = 0; c < C; ++c) {
                const int a = h->cbounds[g * (C + 1) + c], b = h->cbounds[g * (C + 1) + c + 1];
                for (int r = a; r < b; ++r) {
                    chunk_of[(size_t)r] = (int)(g * C + c);
                    first[(size_t)r] = a;
                }
                start[g * C + c + 1] = start[g * C + c] + (size_t)(row_ptr[(size_t)b] - row_ptr[(size_t)a]);
            }
        std::vector<smvp_coo_t> bucket((size_t)std::max(nnz, 1));
        std::vector<size_t> fill(start.begin(), start.end() - 1);
        for (int i = 0; i < nnz; ++i) {
            smvp_coo_t e = coo[i];
            const size_t k = (size_t)chunk_of[(size_t)e.row];
            e.row -= first[(size_t)coo[i].row];
            bucket[fill[k]++] = e;
        }
        // an independent TJDS per row chunk: its output is a disjoint slice of y, so the same all-gather applies
        for (size_t k = 0; rc == SMVP_OK && k < parts; ++k) {
            const size_t g = k / C, c = k % C;
            const int a = h->cbounds[g * (C + 1) + c], b = h->cbounds[g * (C + 1) + c + 1];
            const int pn = (int)(start[k + 1] - start[k]);
            std::vector<int> perm((size_t)std::max(cols, 1)), sp((size_t)std::max(b - a, pn) + 2), ri((size_t)std::max(pn, 1));
            std::vector<double> v((size_t)std::max(pn, 1));
            int nd = 0;
            rc = smvp_tjds_from_coo(bucket.data() + start[k], b - a, cols, pn, perm.data(), sp.data(), (int)sp.size(), ri.data(),
                                    v.data(), &nd, nullptr, nullptr);
            if (rc == SMVP_OK)
                rc = smvp_tjds_create(&h->tjds[k], h->device[g], b - a, cols, pn, nd, perm.data(), sp.data(), ri.data(),
                                      v.data(), SMVP_MEM_HOST);
        }
    }
    if (rc == SMVP_OK && h->auto_exchange)
        rc = smvp_sharded_probe_exchange(h, 3);
    if (rc != SMVP_OK) {
        smvp_sharded_destroy(h);
        return rc;
    }
    *out = h;
    return SMVP_OK;
}

extern "C" int smvp_tjds_sharded_create(smvp_sharded_t **out, int ngpus, const int *devices, const smvp_coo_t *coo,
                                        int rows, int cols, int nnz)
{
    return smvp_tjds_sharded_create_ex(out, ngpus, devices, coo, rows, cols, nnz, nullptr);
}

namespace {

int set_operand_everywhere(smvp_sharded *h, size_t g)
{
    if (h->format != 1)
        return SMVP_OK;
    for (size_t c = 0; c < (size_t)h->chunks; ++c)
        if (int rc = smvp_tjds_set_x(h->tjds[g * (size_t)h->chunks + c], h->d_x[g], h->stream[g]))
            return rc;
    return SMVP_OK;
}

}  // namespace

// CSR blocks: the kernel family of every chunk handle (smvp_csr_set_kernel on each)
extern "C" int smvp_sharded_set_csr_kernel(smvp_sharded_t *h, int kernel, int param)
{
    if (!h || h->format != 0)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_sharded_set_csr_kernel: not a CSR handle");
    for (smvp_csr_t *c : h->csr)
        if (int rc = smvp_csr_set_kernel(c, kernel, param))
            return rc;
    return SMVP_OK;
}

extern "C" int smvp_sharded_set_x(smvp_sharded_t *h, const double *x_host)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    std::vector<double> ones;
    if (!x_host) {  // the reference's operand, main-cli.c:368-369
        ones.assign((size_t)std::max(h->cols, 1), 1.0);
        x_host = ones.data();
    }
    DeviceScope keep;
    for (size_t g = 0; g < (size_t)h->n; ++g) {
        HIP_TRY(hipSetDevice(h->device[g]));
        HIP_TRY(hipMemcpy(h->d_x[g], x_host, sizeof(double) * (size_t)h->cols, hipMemcpyHostToDevice));
        if (int rc = set_operand_everywhere(h, g))
            return rc;
    }
    return SMVP_OK;
}

namespace {

// What GPU g contributes to one product: its chunk products in turn on its compute stream and, by `allgather`, its
// side of the exchange of y.  SMVP_GATHER_OVERLAPPED: the all-gather of chunk c goes onto the communication stream
// right behind the event of chunk c's product -- before chunk c + 1 is even launched -- so chunk c travels while the
// later chunks are multiplied; SMVP_GATHER_AFTER: all gathers on the compute stream after all products (nothing
// overlapped); 0: local products only.  Then one kernel copies the gathered pieces to their rows of the full vector.
// `timed` brackets all of it with an event pair; TJDS chunks that need a cleared y get it first, outside the pair
// (main-cli.c:1008).  Runs on GPU g's issuing thread (the caller's own when there is one GPU): every communicator
// rank is driven by one thread, the ranks' calls come in the same order, so no grouping is needed.
// Host-side meeting of the issuing threads (exchange by copies): returns when all h->n of them have arrived.  A thread
// whose product failed still comes here (issue_product_guarded), so nobody waits for ever.
void phase_barrier(smvp_sharded *h)
{
    if (h->n <= 1)
        return;
    std::unique_lock<std::mutex> lock(h->phase_mu);
    if (h->phase_abort)
        return;
    const unsigned long long round = h->phase_round;
    if (++h->phase_waiting == h->n) {
        h->phase_waiting = 0;
        ++h->phase_round;
        h->phase_cv.notify_all();
    } else {
        h->phase_cv.wait(lock, [&] { return h->phase_abort || h->phase_round != round; });
    }
}

void phase_give_up(smvp_sharded *h)
{
    {
        std::lock_guard<std::mutex> lock(h->phase_mu);
        h->phase_abort = true;
    }
    h->phase_cv.notify_all();
}

int issue_product(smvp_sharded *h, size_t g, int allgather, int timed, bool skip_products)
{
    const size_t C = (size_t)h->chunks;
    const bool overlap = allgather == SMVP_GATHER_OVERLAPPED;
    HIP_TRY(hipSetDevice(h->device[g]));
    if (h->format == 1 && !skip_products)
        for (size_t c = 0; c < C; ++c)
            if (int rc = smvp_tjds_zero_y(h->tjds[g * C + c], h->d_y_local[g] + h->loff[c], h->stream[g]))
                return rc;
    if (timed)
        HIP_TRY(hipEventRecord(h->ev0[g], h->stream[g]));
    const bool pushes = h->exchange != SMVP_EXCHANGE_RCCL;  // COPIES / DIRECT: straight into every rank's full vector
    const size_t n = (size_t)h->n;
    bool placed_awaited[2] = {false, false};  // per stream the pushes go on: [0] the compute stream, [1] the exchange stream
    auto gather = [&](size_t c, hipStream_t st) -> int {
        if (pushes) {  // this rank's chunk c into rows [a, b) of every rank's full vector
            if (!placed_awaited[st == h->stream[g] ? 0 : 1]) {
                // ... once every rank is done with the previous product's full vector (ev_placed: recorded by all of them
                // before anybody left that product, and again behind a feed-back's copy of it)
                for (size_t r = 0; r < n; ++r)
                    HIP_TRY(hipStreamWaitEvent(st, h->ev_placed[r], 0));
                placed_awaited[st == h->stream[g] ? 0 : 1] = true;
            }
            const int a = h->cbounds[g * (C + 1) + c], b = h->cbounds[g * (C + 1) + c + 1];
            if (b <= a)
                return SMVP_OK;
            const double *src = h->d_y_local[g] + h->loff[c];
            if (h->exchange == SMVP_EXCHANGE_COPIES) {
                for (size_t r = 0; r < n; ++r)
                    HIP_TRY(hipMemcpyAsync(h->d_y_full[r] + a, src, sizeof(double) * (size_t)(b - a), hipMemcpyDeviceToDevice, st));
            } else {
                const unsigned bx = (unsigned)std::min(256, std::max(1, (b - a + 4095) / 4096));
                hipLaunchKernelGGL(push_chunk, dim3(bx, (unsigned)n), dim3(256), 0, st, src, h->targets, (long long)a, b - a);
                HIP_TRY(hipGetLastError());
            }
            return SMVP_OK;
        }
        const ncclResult_t nr = h->rccl->AllGather(h->d_y_local[g] + h->loff[c], h->d_wire[g] + h->woff[c], (size_t)h->pad[c],
                                                   ncclDouble, h->comm[g], st);
        if (nr != ncclSuccess)
            return smvp::fail(SMVP_ERR_HIP, "ncclAllGather failed: %s", h->rccl->GetErrorString(nr));
        return SMVP_OK;
    };
    for (size_t c = 0; c < C; ++c) {
        double *yc = h->d_y_local[g] + h->loff[c];
        if (!skip_products) {
            const int rc = h->format == 0 ? smvp_csr_spmv(h->csr[g * C + c], h->d_x[g], yc, h->stream[g])
                                          : smvp_tjds_spmv(h->tjds[g * C + c], yc, h->stream[g]);
            if (rc != SMVP_OK)
                return rc;
        }
        if (overlap) {
            HIP_TRY(hipEventRecord(h->ev_chunk[g * C + c], h->stream[g]));
            HIP_TRY(hipStreamWaitEvent(h->comm_stream[g], h->ev_chunk[g * C + c], 0));
            if (int grc = gather(c, h->comm_stream[g]))
                return grc;
        }
    }
    if (allgather) {
        if (!overlap)
            for (size_t c = 0; c < C; ++c)
                if (int grc = gather(c, h->stream[g]))
                    return grc;
        if (overlap) {
            HIP_TRY(hipEventRecord(h->ev_gathered[g], h->comm_stream[g]));
            HIP_TRY(hipStreamWaitEvent(h->stream[g], h->ev_gathered[g], 0));
        }
        if (pushes) {
            // every rank's chunks must have landed here before this rank's product counts as done: each rank records "pushed"
            // behind its own pushes, the issuing threads meet (so that every event has been recorded), then each waits for all
            HIP_TRY(hipEventRecord(h->ev_pushed[g], h->stream[g]));
            phase_barrier(h);
            for (size_t r = 0; r < n; ++r)
                HIP_TRY(hipStreamWaitEvent(h->stream[g], h->ev_pushed[r], 0));
            HIP_TRY(hipEventRecord(h->ev_placed[g], h->stream[g]));
            phase_barrier(h);  // the next product's pushes wait for these events: all recorded before anybody goes on
        } else if (h->rows > 0) {
            hipLaunchKernelGGL(place_gathered, dim3((unsigned)((h->rows + 255) / 256)), dim3(256), 0, h->stream[g], h->d_wire[g],
                               h->d_y_full[g], h->d_seg[g], h->nseg, h->rows);
            HIP_TRY(hipGetLastError());
        }
    }
    if (timed)
        HIP_TRY(hipEventRecord(h->ev1[g], h->stream[g]));
    return SMVP_OK;
}

void issuer_main(smvp_sharded *h, size_t g)
{
    unsigned long long seen = 0;
    for (;;) {
        int allgather, timed, skip;
        {
            std::unique_lock<std::mutex> lock(h->mu);
            h->cv_go.wait(lock, [&] { return h->quit || h->job != seen; });
            if (h->quit)
                return;
            seen = h->job;
            allgather = h->job_allgather, timed = h->job_timed, skip = h->job_skip_products;
        }
        const int rc = issue_product(h, g, allgather, timed, skip != 0);
        if (rc != SMVP_OK)
            phase_give_up(h);  // (exchange by copies) the peers must not wait for this rank at the meeting points
        {
            std::lock_guard<std::mutex> lock(h->mu);
            h->job_rc[g] = rc;
            h->job_err[g] = rc == SMVP_OK ? "" : smvp_last_error();
            --h->pending;
        }
        h->cv_done.notify_all();
    }
}

}  // namespace

namespace {

// One job on every rank: a product (or, skip_products, only its exchange) enqueued by the ranks' issuing threads.
int run_job(smvp_sharded *h, int allgather, int timed, bool skip_products)
{
    const size_t n = (size_t)h->n;
    if (n == 1 && h->issuer.empty() && smvp::option("sharded_threads", 0) == 0) {  // (plan option: 1 = the issuing threads with one GPU too)
        DeviceScope keep;
        const int rc = issue_product(h, 0, allgather, timed, skip_products);
        if (rc != SMVP_OK && allgather) {
            h->broken = true;
            h->broken_why = smvp_last_error();
        }
        return rc;
    }
    if (h->issuer.empty()) {
        h->job_rc.assign(n, SMVP_OK);
        h->job_err.assign(n, "");
        for (size_t g = 0; g < n; ++g)
            h->issuer.emplace_back(issuer_main, h, g);
    }
    {
        std::unique_lock<std::mutex> lock(h->mu);
        h->job_allgather = allgather, h->job_timed = timed, h->job_skip_products = skip_products ? 1 : 0;
        h->pending = (int)n;
        ++h->job;
        h->cv_go.notify_all();
        h->cv_done.wait(lock, [&] { return h->pending == 0; });
    }
    for (size_t g = 0; g < n; ++g)
        if (h->job_rc[g] != SMVP_OK) {
            // The other ranks have enqueued (or will never get) their side of this product's collectives: those may never
            // complete.  The handle is marked broken so that the error surfaces here and synchronize / destroy do not wait
            // on the device for them (destroy aborts the communicators).
            h->broken = true;
            h->broken_why = "GPU " + std::to_string(h->device[g]) + ": " + h->job_err[g];
            return smvp::fail(h->job_rc[g], "GPU %d: %s", h->device[g], h->job_err[g].c_str());
        }
    return SMVP_OK;
}

bool exchange_available(const smvp_sharded *h, int e)
{
    if (e < SMVP_EXCHANGE_RCCL || e > SMVP_EXCHANGE_DIRECT || h->form_off[e])
        return false;
    return e == SMVP_EXCHANGE_RCCL ? h->have_rccl : h->have_peer;
}

// A form failed while it was being probed (nothing but the probe's own exchanges was in flight): it is taken out of the
// offer and the handle made usable again, so that the probe can go on with the forms that are left.  RCCL: its communicators
// are aborted -- a collective some rank never joined would otherwise keep that rank's stream busy for ever -- which needs
// ncclCommAbort; without it the failure stands.  Returns whether the handle is usable again.
bool retire_form(smvp_sharded *h, int e)
{
    if (e == SMVP_EXCHANGE_RCCL) {
        if (!h->rccl || !h->rccl->CommAbort)
            return false;
        DeviceScope keep;
        for (size_t i = 0; i < h->comm.size(); ++i)
            if (h->comm[i]) {
                (void)hipSetDevice(h->device[i]);
                h->rccl->CommAbort(h->comm[i]);
                h->comm[i] = nullptr;
            }
        h->have_rccl = false;
        h->rccl_why = "failed in the exchange probe: " + std::string(smvp_last_error());
    }
    h->form_off[e] = true;
    {
        DeviceScope keep;
        for (size_t i = 0; i < h->device.size(); ++i) {  // whatever the failed exchange left on the devices has to be over
            if (hipSetDevice(h->device[i]) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
        }
    }
    {
        std::lock_guard<std::mutex> lock(h->phase_mu);
        h->phase_abort = false;
        h->phase_waiting = 0;
    }
    h->broken = false;
    h->broken_why.clear();
    return true;
}

}  // namespace

// One product, asynchronous (see issue_product); returns when every GPU's launches and collectives are enqueued.
extern "C" int smvp_sharded_spmv(smvp_sharded_t *h, int allgather, int timed)
{
    if (!h || allgather < 0 || allgather > SMVP_GATHER_AFTER)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_sharded_spmv: bad argument");
    if (h->broken)
        return smvp::fail(SMVP_ERR_HIP, "the sharded handle is unusable after a failed product (%s): destroy it", h->broken_why.c_str());
    return run_job(h, allgather, timed, false);
}

// The exchange of one product's y -- every chunk of every rank, nothing to overlap with -- under every available form,
// `reps` timed runs behind one untimed; a handle created with SMVP_EXCHANGE_AUTO then keeps the fastest.  What travels is
// whatever y_local holds (zeros right after creation): the bytes and the calls are those of a product.
extern "C" int smvp_sharded_probe_exchange(smvp_sharded_t *h, int reps)
{
    if (!h || reps < 1 || reps > 1000)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_sharded_probe_exchange: bad argument");
    if (h->broken)
        return smvp::fail(SMVP_ERR_HIP, "the sharded handle is unusable after a failed product (%s): destroy it", h->broken_why.c_str());
    const int before = h->exchange;
    int best = -1;
    for (int e = SMVP_EXCHANGE_RCCL; e <= SMVP_EXCHANGE_DIRECT; ++e) {
        h->probe_ms[e] = -1.0;
        if (!exchange_available(h, e))
            continue;
        h->exchange = e;
        double sum = 0.0;
        bool failed = false;
        for (int i = 0; i <= reps; ++i) {
            int rc = run_job(h, SMVP_GATHER_AFTER, 1, true);
            double ms = 0.0;
            if (rc == SMVP_OK)
                rc = smvp_sharded_synchronize(h, &ms);
            if (rc != SMVP_OK) {
                // this form does not work here: go on with the others if the handle can be made usable again, fail only when
                // none is left (one failing form used to fail the whole create under AUTO)
                failed = true;
                if (!retire_form(h, e)) {
                    h->exchange = before;
                    return rc;
                }
                break;
            }
            if (i > 0)
                sum += ms;
        }
        if (failed)
            continue;
        h->probe_ms[e] = sum / reps;
        if (best < 0 || h->probe_ms[e] < h->probe_ms[best])
            best = e;
    }
    if (best < 0)
        return smvp::fail(SMVP_ERR_HIP, "no exchange form survived the probe (last: %s)", smvp_last_error());
    // AUTO leaves its starting form (RCCL where there is one, else the push kernel) only for a form that is more than 5 %
    // faster: three timed exchanges of a few tens of microseconds differ by that much from run to run
    int keep = exchange_available(h, before) ? before : best;
    if (h->auto_exchange && best != keep && !(h->probe_ms[best] < 0.95 * h->probe_ms[keep]))
        best = keep;
    h->exchange = h->auto_exchange ? best : keep;
    return SMVP_OK;
}

extern "C" int smvp_sharded_exchange_info(const smvp_sharded_t *h, int *active, int *available, double *ms, int *rccl_ranks)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (active)
        *active = h->exchange;
    if (available)
        *available = (exchange_available(h, SMVP_EXCHANGE_RCCL) ? 1 << SMVP_EXCHANGE_RCCL : 0) |
                     (exchange_available(h, SMVP_EXCHANGE_COPIES) ? 1 << SMVP_EXCHANGE_COPIES : 0) |
                     (exchange_available(h, SMVP_EXCHANGE_DIRECT) ? 1 << SMVP_EXCHANGE_DIRECT : 0);
    if (ms)
        for (int e = 0; e < 3; ++e)
            ms[e] = h->probe_ms[e];
    if (rccl_ranks) {
        *rccl_ranks = 0;
        if (h->have_rccl && !h->comm.empty() && h->comm[0]) {
            int count = 0;
            if (h->rccl->CommCount && h->rccl->CommCount(h->comm[0], &count) == ncclSuccess)
                *rccl_ranks = count;
            else
                *rccl_ranks = (int)h->comm.size();
        }
    }
    return SMVP_OK;
}

extern "C" int smvp_sharded_set_exchange(smvp_sharded_t *h, int exchange)
{
    if (!h || exchange < SMVP_EXCHANGE_RCCL || exchange > SMVP_EXCHANGE_DIRECT)
        return smvp::fail(SMVP_ERR_INVALID, "smvp_sharded_set_exchange: one of RCCL, COPIES, DIRECT");
    if (!exchange_available(h, exchange))
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "exchange %d is not available on this handle%s%s", exchange,
                          exchange == SMVP_EXCHANGE_RCCL && !h->rccl_why.empty() ? ": " : "",
                          exchange == SMVP_EXCHANGE_RCCL ? h->rccl_why.c_str() : "");
    // products in flight were enqueued under the old form: let them finish before the next one is issued differently
    if (int rc = smvp_sharded_synchronize(h, nullptr))
        return rc;
    h->exchange = exchange;
    return SMVP_OK;
}

// Power iteration: on every GPU the gathered y (divided by its largest magnitude if asked) becomes x.  Every
// GPU normalises its own full copy -- same data, same arithmetic, same result -- so no further exchange is needed.
extern "C" int smvp_sharded_feed_back(smvp_sharded_t *h, int normalize)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (h->rows != h->cols)
        return smvp::fail(SMVP_ERR_INVALID, "power iteration needs a square matrix");
    DeviceScope keep;
    for (size_t g = 0; g < (size_t)h->n; ++g) {
        HIP_TRY(hipSetDevice(h->device[g]));
        if (normalize)
            HIP_TRY(smvp::launch_normalize_max(h->d_y_full[g], h->rows, h->d_norm[g], h->stream[g]));
        HIP_TRY(hipMemcpyAsync(h->d_x[g], h->d_y_full[g], sizeof(double) * (size_t)h->rows, hipMemcpyDeviceToDevice,
                               h->stream[g]));
        // the push forms write the next product's chunks straight into the full vectors: not before this copy has read them
        HIP_TRY(hipEventRecord(h->ev_placed[g], h->stream[g]));
        if (int rc = set_operand_everywhere(h, g))
            return rc;
    }
    return SMVP_OK;
}

// Waits for every GPU; *ms (optional) = the longest event-pair time of the last timed product.
extern "C" int smvp_sharded_synchronize(smvp_sharded_t *h, double *ms)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (h->broken)
        return smvp::fail(SMVP_ERR_HIP, "the sharded handle is unusable after a failed product (%s): destroy it", h->broken_why.c_str());
    DeviceScope keep;
    double worst = 0.0;
    for (size_t g = 0; g < (size_t)h->n; ++g) {
        HIP_TRY(hipSetDevice(h->device[g]));
        HIP_TRY(hipStreamSynchronize(h->comm_stream[g]));
        HIP_TRY(hipStreamSynchronize(h->stream[g]));
        if (ms) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, h->ev0[g], h->ev1[g]) == hipSuccess)
                worst = std::max(worst, (double)t);
        }
    }
    if (ms)
        *ms = worst;
    return SMVP_OK;
}

// y[rows] from GPU `slot`: its gathered full vector (gathered != 0) or the slices collected from every GPU.
extern "C" int smvp_sharded_get_y(smvp_sharded_t *h, int slot, int gathered, double *y_host)
{
    if (!h || slot < 0 || slot >= h->n || (h->rows > 0 && !y_host))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_sharded_get_y: bad argument");
    DeviceScope keep;
    if (gathered) {
        HIP_TRY(hipSetDevice(h->device[(size_t)slot]));
        if (h->rows > 0)
            HIP_TRY(hipMemcpy(y_host, h->d_y_full[(size_t)slot], sizeof(double) * (size_t)h->rows, hipMemcpyDeviceToHost));
        return SMVP_OK;
    }
    const size_t C = (size_t)h->chunks;
    for (size_t g = 0; g < (size_t)h->n; ++g) {
        HIP_TRY(hipSetDevice(h->device[g]));
        for (size_t c = 0; c < C; ++c) {
            const int a = h->cbounds[g * (C + 1) + c], b = h->cbounds[g * (C + 1) + c + 1];
            if (b > a)
                HIP_TRY(hipMemcpy(y_host + a, h->d_y_local[g] + h->loff[c], sizeof(double) * (size_t)(b - a), hipMemcpyDeviceToHost));
        }
    }
    return SMVP_OK;
}

extern "C" int smvp_sharded_info(const smvp_sharded_t *h, int *ngpus, int *rows_per_gpu)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (ngpus)
        *ngpus = h->n;
    if (rows_per_gpu) {  // the tallest block
        int tallest = 0;
        for (int g = 0; g < h->n; ++g)
            tallest = std::max(tallest, h->bounds[(size_t)g + 1] - h->bounds[(size_t)g]);
        *rows_per_gpu = tallest;
    }
    return SMVP_OK;
}

// bounds[ngpus + 1] of the row blocks and, if asked for, chunk_bounds[ngpus * (chunks + 1)] of their chunks (global rows)
extern "C" int smvp_sharded_layout(const smvp_sharded_t *h, int *chunks, int *bounds, int *chunk_bounds)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (chunks)
        *chunks = h->chunks;
    if (bounds)
        memcpy(bounds, h->bounds.data(), sizeof(int) * h->bounds.size());
    if (chunk_bounds)
        memcpy(chunk_bounds, h->cbounds.data(), sizeof(int) * h->cbounds.size());
    return SMVP_OK;
}
