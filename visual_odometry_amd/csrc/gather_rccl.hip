// gather_rccl.hip — the one exchange step of the path: the trajectory gather.
//
// Frame pairs are independent, so ranks (one process per GPU) share nothing while they detect, match and solve; at the
// end of a batch every rank contributes its per-pair records ([R|t] + counts, 16 float64 = 128 B per pair) to ONE
// ncclAllGather over RCCL / xGMI.  The records are packed on the device straight from the batch's result array
// (vo_pair_result, in HBM) and gathered over the process's ONE communicator on the context's own stream, each collective
// ordered behind the previous one with an event — no host bounce, no torch tensors.  RCCL is
// bound at run time (dlopen: the copy already loaded by the process if there is one), so libvo_hip.so has no link
// dependency on it and single-GPU users never load it.
#include "vo_internal.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
};

static RcclApi g_rccl;

const char* rccl_load(void)
{
    if (g_rccl.handle) return nullptr;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);          // share the copy the process already uses
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return "librccl.so.1 not found (dlopen)";
    RcclApi a;
    a.handle = h;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    a.CommCount = (decltype(a.CommCount))dlsym(h, "ncclCommCount");
    if (!a.GetUniqueId || !a.CommCount || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString) return "librccl.so.1 lacks an expected symbol";
    g_rccl = a;
    return nullptr;
}

const char* rccl_unique_id(uint8_t* id128)
{
    const char* e = rccl_load();
    if (e) return e;
    ncclUniqueId id;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return g_rccl.GetErrorString(r);
    static_assert(sizeof(id) == VO_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof(id));
    return nullptr;
}

const char* rccl_comm_init(void** comm, const uint8_t* id128, int rank, int world)
{
    const char* e = rccl_load();
    if (e) return e;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t r = g_rccl.CommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) return g_rccl.GetErrorString(r);
    *comm = (void*)c;
    return nullptr;
}

void rccl_comm_destroy(void* comm)
{
    if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy((ncclComm_t)comm);
}

const char* rccl_comm_count(void* comm, int* n)
{
    ncclResult_t r = g_rccl.CommCount((ncclComm_t)comm, n);
    return r == ncclSuccess ? nullptr : g_rccl.GetErrorString(r);
}

const char* rccl_all_gather_f64(void* comm, const double* send, double* recv, size_t count, hipStream_t s)
{
    ncclResult_t r = g_rccl.AllGather(send, recv, count, ncclFloat64, (ncclComm_t)comm, s);
    return r == ncclSuccess ? nullptr : g_rccl.GetErrorString(r);
}

// vo_pair_result -> 16 float64: R (9), t (3), n_kp1, n_match, n_inl, n_good (pairs that failed keep their status as a
// negative n_inl so the receiver can tell)
// rows at and beyond `valid` (the pair count of the run that filled `res`) are padding of a short or empty block: zeros with
// VO_ERR_NOT_CONFIGURED in the n_inl column, never whatever an earlier batch left in the buffer
__global__ void k_pack_records(const vo_pair_result* res, int B, int valid, double* rec)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * VO_RECORD_DOUBLES) return;
    const int p = i / VO_RECORD_DOUBLES, k = i % VO_RECORD_DOUBLES;
    if (p >= valid) { rec[i] = k == 14 ? (double)VO_ERR_NOT_CONFIGURED : 0.0; return; }
    const vo_pair_result& r = res[p];
    double v;
    if (k < 9) v = r.R[k];
    else if (k < 12) v = r.t[k - 9];
    else if (k == 12) v = (double)r.n_kp1;
    else if (k == 13) v = (double)r.n_match;
    else if (k == 14) v = r.status == VO_OK ? (double)r.n_inl : (double)r.status;
    else v = (double)r.n_good;
    rec[i] = v;
}

void launch_pack_records(hipStream_t s, const vo_pair_result* res, int B, int valid, double* rec)
{
    if (B <= 0) return;
    hipLaunchKernelGGL(k_pack_records, dim3((B * VO_RECORD_DOUBLES + 255) / 256), dim3(256), 0, s, res, B, valid, rec);
}
