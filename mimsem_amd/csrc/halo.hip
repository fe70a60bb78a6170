// mimsem_amd/csrc/halo.hip -- the device-side halo exchange of the C ABI: mimsem_halo_create / _begin / _end
// (replaces VecScatterBegin/End on gtol_0 / gtol_1, eul/Topo.cpp:145-155, eul/Assembly.cpp:2194-2195, eul/Euler_2.cpp:1455-1456).
//
// A C++ host (north_star: "host code stays C++") gets the whole exchange without Python and without staging through PCIe:
//   begin: ONE pack launch for all neighbours (k_halo_segments) on the context's stream -> event -> the TRANSPORT on a separate
//          communication stream (so interior work enqueued between begin and end overlaps the exchange);
//   end  : the context's stream waits for the transport's event -> unpack (INSERT, or ADD in rank-ordered ranges so that sums
//          whose targets repeat between neighbours are formed in a fixed order).
// Transports: RCCL (grouped ncclSend/ncclRecv over xGMI; librccl is dlopen'ed on first use -- the process's already-loaded copy when
// there is one --, the communicator is the host's), a host callback (GPU-aware MPI, torch.distributed, a test double), or loop-back
// (a plan whose only neighbour is the rank itself: periodic single-rank layouts and tests).
#include <dlfcn.h>
#include <mutex>
#include <set>
#include <vector>
#include "ctx.hpp"

struct mimsem_halo {
    mimsem_ctx* c = nullptr;
    int nneigh = 0, max_nlev = 0;
    std::vector<int> ranks, send_off, recv_off;              // offsets in slots (per level), prefix sums, host
    std::vector<std::pair<int, int>> add_ranges;             // neighbour ranges whose receive slots are pairwise disjoint
    int *d_send_idx = nullptr, *d_recv_idx = nullptr;
    double *d_send = nullptr, *d_recv = nullptr;
    hipStream_t comm = nullptr;
    hipEvent_t ev_packed = nullptr, ev_done = nullptr;
    int transport = 0;                                       // 0 none, 1 callback, 2 loop-back, 3 RCCL
    mimsem_halo_transport_fn fn = nullptr; void* user = nullptr;
    void* nccl_comm = nullptr;
    // state of the exchange in flight
    bool in_flight = false; int mode = 0, nlev = 0; double* v = nullptr; long long vs = 0;
};

namespace {
// the four RCCL entry points the built-in transport needs, resolved at run time (no link-time dependency on librccl)
struct Rccl {
    int (*GroupStart)() = nullptr; int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    bool ok = false;
};
// The communicator handed to mimsem_halo_set_rccl belongs to ONE instance of the library: the host's.  So the entry points are taken
// from (1) the handle the host passed (mimsem_halo_use_rccl_library), else (2) the copy the process has ALREADY loaded
// (RTLD_NOLOAD) -- never from a second copy this library would load by itself (a bundled or renamed librccl would otherwise end up
// with an ncclComm_t created by another instance of it).
std::once_flag g_rccl_once;
void* g_rccl_handle = nullptr;                                        // set before the first use, if at all
Rccl g_rccl;
void rccl_resolve() {
    void* h = g_rccl_handle;
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) return;
    Rccl& r = g_rccl;
    r.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclSend");
    r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclRecv");
    r.ok = r.GroupStart && r.GroupEnd && r.Send && r.Recv;
}
Rccl& rccl() {
    std::call_once(g_rccl_once, rccl_resolve);
    return g_rccl;
}
constexpr int kNcclFloat64 = 8;                                       // ncclDataType_t: ncclDouble
}  // namespace

extern "C" {

int mimsem_halo_create(mimsem_ctx* c, int nneigh, const int* ranks, const int* send_idx, const int* send_off,
                       const int* recv_idx, const int* recv_off, int nslots, int max_nlev, mimsem_halo** out) {
    if (!c || !out || nneigh < 0 || nneigh > MIMSEM_HALO_MAX_SEGMENTS || max_nlev < 1 || nslots < 0) return MIMSEM_ERR_ARG;
    *out = nullptr;
    if (nneigh && (!ranks || !send_off || !recv_off)) return MIMSEM_ERR_ARG;
    if (nneigh && (send_off[0] != 0 || recv_off[0] != 0)) return MIMSEM_ERR_ARG;        // prefix sums start at 0
    for (int i = 0; i < nneigh; i++) if (send_off[i + 1] < send_off[i] || recv_off[i + 1] < recv_off[i]) return MIMSEM_ERR_ARG;
    const int ns = nneigh ? send_off[nneigh] : 0, nr = nneigh ? recv_off[nneigh] : 0;
    if ((ns && !send_idx) || (nr && !recv_idx)) return MIMSEM_ERR_ARG;
    for (int i = 0; i < ns; i++) if (send_idx[i] < 0 || send_idx[i] >= nslots) return MIMSEM_ERR_ARG;
    for (int i = 0; i < nr; i++) if (recv_idx[i] < 0 || recv_idx[i] >= nslots) return MIMSEM_ERR_ARG;
    MIMSEM_HIP_TRY(hipSetDevice(c->device));
    mimsem_halo* h = new mimsem_halo();
    h->c = c; h->nneigh = nneigh; h->max_nlev = max_nlev;
    h->ranks.assign(ranks, ranks + nneigh);
    h->send_off.assign(send_off, send_off + nneigh + 1); h->recv_off.assign(recv_off, recv_off + nneigh + 1);
    if (!nneigh) { h->send_off = {0}; h->recv_off = {0}; }
    // ADD order: greedy ranges of neighbours (in the order given) with pairwise disjoint receive slots.  A slot listed TWICE inside
    // one neighbour's segment cannot be split off that way (k_halo_segments adds a segment's entries concurrently): rejected.
    {
        std::set<int> seen; int start = 0;
        for (int i = 0; i < nneigh; i++) {
            std::set<int> own;
            for (int k = recv_off[i]; k < recv_off[i + 1]; k++) if (!own.insert(recv_idx[k]).second) { delete h; return MIMSEM_ERR_ARG; }
            bool clash = false;
            for (int k = recv_off[i]; k < recv_off[i + 1] && !clash; k++) clash = seen.count(recv_idx[k]) != 0;
            if (clash) { h->add_ranges.push_back({start, i}); start = i; seen.clear(); }
            for (int k = recv_off[i]; k < recv_off[i + 1]; k++) seen.insert(recv_idx[k]);
        }
        if (nneigh) h->add_ranges.push_back({start, nneigh});
    }
    auto fail = [&](hipError_t e, const char* what) { mimsem_halo_destroy(h); return mimsem::hip_fail(e, what); };
    hipError_t e;
    if ((e = hipMalloc((void**)&h->d_send_idx, std::max(ns, 1)*sizeof(int))) != hipSuccess) return fail(e, "hipMalloc(halo)");
    if ((e = hipMalloc((void**)&h->d_recv_idx, std::max(nr, 1)*sizeof(int))) != hipSuccess) return fail(e, "hipMalloc(halo)");
    if ((e = hipMalloc((void**)&h->d_send, (size_t)std::max(ns, 1)*max_nlev*sizeof(double))) != hipSuccess) return fail(e, "hipMalloc(halo)");
    if ((e = hipMalloc((void**)&h->d_recv, (size_t)std::max(nr, 1)*max_nlev*sizeof(double))) != hipSuccess) return fail(e, "hipMalloc(halo)");
    if (ns && (e = hipMemcpy(h->d_send_idx, send_idx, ns*sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(halo)");
    if (nr && (e = hipMemcpy(h->d_recv_idx, recv_idx, nr*sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return fail(e, "hipMemcpy(halo)");
    if ((e = hipStreamCreateWithFlags(&h->comm, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate(halo)");
    if ((e = hipEventCreateWithFlags(&h->ev_packed, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate(halo)");
    if ((e = hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate(halo)");
    *out = h;
    return MIMSEM_OK;
}

void mimsem_halo_destroy(mimsem_halo* h) {
    if (!h) return;
    if (h->comm) { (void)hipStreamSynchronize(h->comm); (void)hipStreamDestroy(h->comm); }
    if (h->ev_packed) (void)hipEventDestroy(h->ev_packed);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    void* ptrs[] = {h->d_send_idx, h->d_recv_idx, h->d_send, h->d_recv};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete h;
}

int mimsem_halo_set_transport(mimsem_halo* h, mimsem_halo_transport_fn fn, void* user) {
    if (!h || !fn || h->in_flight) return MIMSEM_ERR_ARG;
    h->transport = 1; h->fn = fn; h->user = user;
    return MIMSEM_OK;
}
int mimsem_halo_set_loopback(mimsem_halo* h) {
    if (!h || h->in_flight) return MIMSEM_ERR_ARG;
    // every message goes to the rank itself: legal only when send and receive segments match in size
    for (int i = 0; i < h->nneigh; i++)
        if (h->send_off[i + 1] - h->send_off[i] != h->recv_off[i + 1] - h->recv_off[i]) return MIMSEM_ERR_ARG;
    h->transport = 2;
    return MIMSEM_OK;
}
int mimsem_halo_use_rccl_library(void* dl_handle) {
    if (!dl_handle) return MIMSEM_ERR_ARG;
    bool first = false;
    g_rccl_handle = dl_handle;                                     // read once, by the first rccl() below or later
    std::call_once(g_rccl_once, [&] { first = true; rccl_resolve(); });
    if (!first) return MIMSEM_ERR_STATE;                           // the entry points were resolved before: too late to change the library
    return g_rccl.ok ? MIMSEM_OK : MIMSEM_ERR_STATE;
}
int mimsem_halo_set_rccl(mimsem_halo* h, void* nccl_comm) {
    if (!h || !nccl_comm || h->in_flight) return MIMSEM_ERR_ARG;
    if (!rccl().ok) return MIMSEM_ERR_STATE;                       // no librccl loaded in this process, and none handed over
    h->transport = 3; h->nccl_comm = nccl_comm;
    return MIMSEM_OK;
}

int mimsem_halo_begin(mimsem_halo* h, int mode, int nlev, double* v, long long vs) {
    if (!h || !v || nlev < 1 || nlev > h->max_nlev || (mode != MIMSEM_HALO_INSERT && mode != MIMSEM_HALO_ADD)) return MIMSEM_ERR_ARG;
    if (h->in_flight || h->transport == 0) return MIMSEM_ERR_STATE;
    mimsem_ctx* c = h->c;
    int rc;
    if (h->nneigh && h->send_off[h->nneigh]) {
        rc = launch_halo_segments(c, h->d_send_idx, h->nneigh, h->send_off.data(), 0, h->nneigh, nlev, 0, h->d_send, v, vs);
        if (rc) return rc;
    }
    MIMSEM_HIP_TRY(hipEventRecord(h->ev_packed, c->stream));
    MIMSEM_HIP_TRY(hipStreamWaitEvent(h->comm, h->ev_packed, 0));
    if (h->nneigh) {
        if (h->transport == 1) {
            std::vector<long long> so(h->nneigh + 1), ro(h->nneigh + 1);
            for (int i = 0; i <= h->nneigh; i++) { so[i] = (long long)h->send_off[i]*nlev; ro[i] = (long long)h->recv_off[i]*nlev; }
            rc = h->fn(h->user, h->d_send, so.data(), h->d_recv, ro.data(), h->nneigh, h->ranks.data(), (void*)h->comm);
            if (rc) return MIMSEM_ERR_STATE;
        } else if (h->transport == 2) {
            const size_t bytes = (size_t)h->send_off[h->nneigh]*nlev*sizeof(double);
            if (bytes) MIMSEM_HIP_TRY(hipMemcpyAsync(h->d_recv, h->d_send, bytes, hipMemcpyDeviceToDevice, h->comm));
        } else {
            Rccl& r = rccl();
            if (r.GroupStart()) return MIMSEM_ERR_STATE;
            bool failed = false;                                     // a group once opened is ALWAYS closed: an open group would swallow the host's next collectives
            for (int i = 0; i < h->nneigh && !failed; i++) {
                const size_t ns = (size_t)(h->send_off[i + 1] - h->send_off[i])*nlev, nr = (size_t)(h->recv_off[i + 1] - h->recv_off[i])*nlev;
                if (ns && r.Send(h->d_send + (size_t)h->send_off[i]*nlev, ns, kNcclFloat64, h->ranks[i], h->nccl_comm, h->comm)) failed = true;
                if (!failed && nr && r.Recv(h->d_recv + (size_t)h->recv_off[i]*nlev, nr, kNcclFloat64, h->ranks[i], h->nccl_comm, h->comm)) failed = true;
            }
            if (r.GroupEnd() || failed) return MIMSEM_ERR_STATE;
        }
    }
    MIMSEM_HIP_TRY(hipEventRecord(h->ev_done, h->comm));
    h->in_flight = true; h->mode = mode; h->nlev = nlev; h->v = v; h->vs = vs;
    return MIMSEM_OK;
}

int mimsem_halo_end(mimsem_halo* h) {
    if (!h) return MIMSEM_ERR_ARG;
    if (!h->in_flight) return MIMSEM_ERR_STATE;
    mimsem_ctx* c = h->c;
    h->in_flight = false;
    MIMSEM_HIP_TRY(hipStreamWaitEvent(c->stream, h->ev_done, 0));
    if (!h->nneigh || !h->recv_off[h->nneigh]) return MIMSEM_OK;
    if (h->mode == MIMSEM_HALO_INSERT)
        return launch_halo_segments(c, h->d_recv_idx, h->nneigh, h->recv_off.data(), 0, h->nneigh, h->nlev, 1, h->d_recv, h->v, h->vs);
    for (auto& r : h->add_ranges) {                                  // fixed order => reproducible sums
        int rc = launch_halo_segments(c, h->d_recv_idx, h->nneigh, h->recv_off.data(), r.first, r.second, h->nlev, 2, h->d_recv, h->v, h->vs);
        if (rc) return rc;
    }
    return MIMSEM_OK;
}

}  // extern "C"
