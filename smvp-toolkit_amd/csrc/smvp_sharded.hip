// smvp_sharded.hip -- one product over several GPUs of a node from ONE host process.
//
// The reference is a single thread on one CPU; this is new design:
// the matrix is cut into row blocks of equal height, GPU g holds block g (its own
// CSR or TJDS handle), all of x, and produces its slice of y; one RCCL
// ncclAllGather over xGMI assembles the full y on every GPU.  bench.py does the
// same with one process per GPU through torch.distributed; this file is what the
// C command line (--gpus N) and smvp_*_compute(opts.ngpus > 1) use.
//
// librccl is loaded with dlopen the first time more than zero GPUs are sharded,
// so single-GPU runs neither link nor load it.
#include "smvp_common.h"
#include "smvp_kernels.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstring>
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
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int load_rccl(Rccl **out)
{
    static Rccl r;
    static int status = -1;
    if (status < 0) {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib)
                break;
        }
        status = SMVP_ERR_UNSUPPORTED;
        if (r.lib) {
            r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
            r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
            r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
            r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
            if (r.CommInitAll && r.CommDestroy && r.AllGather && r.GroupStart && r.GroupEnd && r.GetErrorString)
                status = SMVP_OK;
        }
    }
    if (status != SMVP_OK)
        return smvp::fail(SMVP_ERR_UNSUPPORTED, "RCCL (librccl.so) could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
    *out = &r;
    return SMVP_OK;
}

}  // namespace

struct smvp_sharded {
    int format = 0;  // 0 = CSR, 1 = TJDS
    int n = 0;
    int rows = 0, cols = 0, nnz = 0, block = 0;  // block = rows per GPU (the last one may hold fewer)
    std::vector<int> device, r0, r1;
    std::vector<smvp_csr_t *> csr;
    std::vector<smvp_tjds_t *> tjds;
    std::vector<hipStream_t> stream;
    std::vector<double *> d_x, d_y_local, d_y_full;
    std::vector<hipEvent_t> ev0, ev1;
    std::vector<ncclComm_t> comm;
    std::vector<unsigned long long *> d_norm;
    Rccl *rccl = nullptr;
};

extern "C" void smvp_sharded_destroy(smvp_sharded_t *h)
{
    if (!h)
        return;
    for (int g = 0; g < (int)h->device.size(); ++g) {
        (void)hipSetDevice(h->device[(size_t)g]);
        if (g < (int)h->comm.size() && h->comm[(size_t)g] && h->rccl)
            h->rccl->CommDestroy(h->comm[(size_t)g]);
        if (g < (int)h->csr.size())
            smvp_csr_destroy(h->csr[(size_t)g]);
        if (g < (int)h->tjds.size())
            smvp_tjds_destroy(h->tjds[(size_t)g]);
        for (auto *vec : {&h->d_x, &h->d_y_local, &h->d_y_full})
            if (g < (int)vec->size() && (*vec)[(size_t)g])
                (void)hipFree((*vec)[(size_t)g]);
        if (g < (int)h->d_norm.size() && h->d_norm[(size_t)g])
            (void)hipFree(h->d_norm[(size_t)g]);
        if (g < (int)h->ev0.size() && h->ev0[(size_t)g])
            (void)hipEventDestroy(h->ev0[(size_t)g]);
        if (g < (int)h->ev1.size() && h->ev1[(size_t)g])
            (void)hipEventDestroy(h->ev1[(size_t)g]);
        if (g < (int)h->stream.size() && h->stream[(size_t)g])
            (void)hipStreamDestroy(h->stream[(size_t)g]);
    }
    delete h;
}

namespace {

int sharded_common(smvp_sharded *h, int ngpus, const int *devices, int rows, int cols, int nnz)
{
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
        return smvp::fail(SMVP_ERR_NO_DEVICE, "no HIP device is visible (this engine has no CPU path)");
    if (ngpus < 1 || ngpus > visible)
        return smvp::fail(SMVP_ERR_INVALID, "%d GPUs requested, %d visible", ngpus, visible);
    h->n = ngpus;
    h->rows = rows, h->cols = cols, h->nnz = nnz;
    h->block = std::max(1, (rows + ngpus - 1) / ngpus);
    for (int g = 0; g < ngpus; ++g) {
        const int dev = devices ? devices[g] : g;
        if (dev < 0 || dev >= visible)
            return smvp::fail(SMVP_ERR_INVALID, "device %d out of range", dev);
        for (int p : h->device)
            if (p == dev)
                return smvp::fail(SMVP_ERR_INVALID, "device %d listed twice", dev);
        h->device.push_back(dev);
        h->r0.push_back(std::min(rows, g * h->block));
        h->r1.push_back(std::min(rows, (g + 1) * h->block));
    }
    h->stream.assign((size_t)ngpus, nullptr);
    h->d_x.assign((size_t)ngpus, nullptr);
    h->d_y_local.assign((size_t)ngpus, nullptr);
    h->d_y_full.assign((size_t)ngpus, nullptr);
    h->ev0.assign((size_t)ngpus, nullptr);
    h->ev1.assign((size_t)ngpus, nullptr);
    h->d_norm.assign((size_t)ngpus, nullptr);
    for (int g = 0; g < ngpus; ++g) {
        HIP_TRY(hipSetDevice(h->device[(size_t)g]));
        HIP_TRY(hipStreamCreate(&h->stream[(size_t)g]));
        HIP_TRY(hipEventCreate(&h->ev0[(size_t)g]));
        HIP_TRY(hipEventCreate(&h->ev1[(size_t)g]));
        HIP_TRY(hipMalloc((void **)&h->d_x[(size_t)g], sizeof(double) * (size_t)std::max(std::max(cols, rows), 1)));
        HIP_TRY(hipMalloc((void **)&h->d_norm[(size_t)g], sizeof(unsigned long long)));
        HIP_TRY(hipMalloc((void **)&h->d_y_local[(size_t)g], sizeof(double) * (size_t)h->block));
        HIP_TRY(hipMalloc((void **)&h->d_y_full[(size_t)g], sizeof(double) * (size_t)h->block * (size_t)ngpus));
        HIP_TRY(hipMemset(h->d_y_local[(size_t)g], 0, sizeof(double) * (size_t)h->block));
        HIP_TRY(hipMemset(h->d_y_full[(size_t)g], 0, sizeof(double) * (size_t)h->block * (size_t)ngpus));
    }
    if (int rc = load_rccl(&h->rccl))
        return rc;
    h->comm.assign((size_t)ngpus, nullptr);
    ncclResult_t nr = h->rccl->CommInitAll(h->comm.data(), ngpus, h->device.data());
    if (nr != ncclSuccess)
        return smvp::fail(SMVP_ERR_HIP, "ncclCommInitAll failed: %s", h->rccl->GetErrorString(nr));
    return SMVP_OK;
}

}  // namespace

extern "C" int smvp_csr_sharded_create(smvp_sharded_t **out, int ngpus, const int *devices, int rows, int cols, int nnz,
                                       const int *row_ptr, const int *col_ind, const double *val)
{
    if (!out || rows < 0 || cols < 0 || nnz < 0 || !row_ptr || (nnz > 0 && (!col_ind || !val)))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_csr_sharded_create: bad argument");
    smvp_sharded *h = new smvp_sharded;
    h->format = 0;
    int rc = sharded_common(h, ngpus, devices, rows, cols, nnz);
    h->csr.assign((size_t)std::max(ngpus, 0), nullptr);
    for (int g = 0; rc == SMVP_OK && g < ngpus; ++g) {
        const int a = h->r0[(size_t)g], b = h->r1[(size_t)g];
        std::vector<int> rp((size_t)(b - a) + 1);
        for (int r = a; r <= b; ++r)
            rp[(size_t)(r - a)] = row_ptr[r] - row_ptr[a];
        rc = smvp_csr_create(&h->csr[(size_t)g], h->device[(size_t)g], b - a, cols, row_ptr[b] - row_ptr[a], rp.data(),
                             col_ind + row_ptr[a], val + row_ptr[a], SMVP_MEM_HOST, nullptr);
    }
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
    if (!out || rows < 0 || cols < 0 || nnz < 0 || (nnz > 0 && !coo))
        return smvp::fail(SMVP_ERR_INVALID, "smvp_tjds_sharded_create: bad argument");
    smvp_sharded *h = new smvp_sharded;
    h->format = 1;
    int rc = sharded_common(h, ngpus, devices, rows, cols, nnz);
    h->tjds.assign((size_t)std::max(ngpus, 0), nullptr);
    // an independent TJDS per row block: its output is a disjoint slice of y, so the same all-gather applies
    for (int g = 0; rc == SMVP_OK && g < ngpus; ++g) {
        const int a = h->r0[(size_t)g], b = h->r1[(size_t)g];
        std::vector<smvp_coo_t> part;
        for (int i = 0; i < nnz; ++i)
            if (coo[i].row >= a && coo[i].row < b) {
                part.push_back(coo[i]);
                part.back().row -= a;
            }
        const int pn = (int)part.size();
        std::vector<int> perm((size_t)std::max(cols, 1)), sp((size_t)std::max(b - a, pn) + 2), ri((size_t)std::max(pn, 1));
        std::vector<double> v((size_t)std::max(pn, 1));
        int nd = 0;
        rc = smvp_tjds_from_coo(part.data(), b - a, cols, pn, perm.data(), sp.data(), (int)sp.size(), ri.data(), v.data(),
                                &nd, nullptr, nullptr);
        if (rc == SMVP_OK)
            rc = smvp_tjds_create(&h->tjds[(size_t)g], h->device[(size_t)g], b - a, cols, pn, nd, perm.data(), sp.data(),
                                  ri.data(), v.data(), SMVP_MEM_HOST);
    }
    if (rc != SMVP_OK) {
        smvp_sharded_destroy(h);
        return rc;
    }
    *out = h;
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
    for (int g = 0; g < h->n; ++g) {
        HIP_TRY(hipSetDevice(h->device[(size_t)g]));
        HIP_TRY(hipMemcpy(h->d_x[(size_t)g], x_host, sizeof(double) * (size_t)h->cols, hipMemcpyHostToDevice));
        if (h->format == 1)
            if (int rc = smvp_tjds_set_x(h->tjds[(size_t)g], h->d_x[(size_t)g], h->stream[(size_t)g]))
                return rc;
    }
    return SMVP_OK;
}

// One product: local products on every GPU, then (allgather != 0) one grouped ncclAllGather.  `timed` records the
// event pair on every GPU around exactly that; TJDS blocks clear their y slice first, outside the pair.
extern "C" int smvp_sharded_spmv(smvp_sharded_t *h, int allgather, int timed)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    for (int g = 0; g < h->n; ++g) {
        const size_t i = (size_t)g;
        HIP_TRY(hipSetDevice(h->device[i]));
        if (h->format == 1)
            if (int rc = smvp_tjds_zero_y(h->tjds[i], h->d_y_local[i], h->stream[i]))
                return rc;
        if (timed)
            HIP_TRY(hipEventRecord(h->ev0[i], h->stream[i]));
        const int rc = h->format == 0 ? smvp_csr_spmv(h->csr[i], h->d_x[i], h->d_y_local[i], h->stream[i])
                                      : smvp_tjds_spmv(h->tjds[i], h->d_y_local[i], h->stream[i]);
        if (rc != SMVP_OK)
            return rc;
    }
    if (allgather) {
        ncclResult_t nr = h->rccl->GroupStart();
        for (int g = 0; g < h->n && nr == ncclSuccess; ++g) {
            const size_t i = (size_t)g;
            nr = h->rccl->AllGather(h->d_y_local[i], h->d_y_full[i], (size_t)h->block, ncclDouble, h->comm[i], h->stream[i]);
        }
        const ncclResult_t ne = h->rccl->GroupEnd();
        if (nr == ncclSuccess)
            nr = ne;
        if (nr != ncclSuccess)
            return smvp::fail(SMVP_ERR_HIP, "ncclAllGather failed: %s", h->rccl->GetErrorString(nr));
    }
    if (timed)
        for (int g = 0; g < h->n; ++g) {
            HIP_TRY(hipSetDevice(h->device[(size_t)g]));
            HIP_TRY(hipEventRecord(h->ev1[(size_t)g], h->stream[(size_t)g]));
        }
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
    for (int g = 0; g < h->n; ++g) {
        const size_t i = (size_t)g;
        HIP_TRY(hipSetDevice(h->device[i]));
        if (normalize)
            HIP_TRY(smvp::launch_normalize_max(h->d_y_full[i], h->rows, h->d_norm[i], h->stream[i]));
        HIP_TRY(hipMemcpyAsync(h->d_x[i], h->d_y_full[i], sizeof(double) * (size_t)h->rows, hipMemcpyDeviceToDevice,
                               h->stream[i]));
        if (h->format == 1)
            if (int rc = smvp_tjds_set_x(h->tjds[i], h->d_x[i], h->stream[i]))
                return rc;
    }
    return SMVP_OK;
}

// Waits for every GPU; *ms (optional) = the longest event-pair time of the last timed product.
extern "C" int smvp_sharded_synchronize(smvp_sharded_t *h, double *ms)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    double worst = 0.0;
    for (int g = 0; g < h->n; ++g) {
        HIP_TRY(hipSetDevice(h->device[(size_t)g]));
        HIP_TRY(hipStreamSynchronize(h->stream[(size_t)g]));
        if (ms) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, h->ev0[(size_t)g], h->ev1[(size_t)g]) == hipSuccess)
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
    if (gathered) {
        HIP_TRY(hipSetDevice(h->device[(size_t)slot]));
        if (h->rows > 0)
            HIP_TRY(hipMemcpy(y_host, h->d_y_full[(size_t)slot], sizeof(double) * (size_t)h->rows, hipMemcpyDeviceToHost));
        return SMVP_OK;
    }
    for (int g = 0; g < h->n; ++g) {
        const int a = h->r0[(size_t)g], b = h->r1[(size_t)g];
        HIP_TRY(hipSetDevice(h->device[(size_t)g]));
        if (b > a)
            HIP_TRY(hipMemcpy(y_host + a, h->d_y_local[(size_t)g], sizeof(double) * (size_t)(b - a), hipMemcpyDeviceToHost));
    }
    return SMVP_OK;
}

extern "C" int smvp_sharded_info(const smvp_sharded_t *h, int *ngpus, int *rows_per_gpu)
{
    if (!h)
        return smvp::fail(SMVP_ERR_INVALID, "null handle");
    if (ngpus)
        *ngpus = h->n;
    if (rows_per_gpu)
        *rows_per_gpu = h->block;
    return SMVP_OK;
}
