// mimsem_amd/csrc/halo.hip -- the device-side halo exchange of the C ABI: mimsem_halo_create / _begin / _end
// (replaces VecScatterBegin/End on gtol_0 / gtol_1, eul/Topo.cpp:145-155, eul/Assembly.cpp:2194-2195, eul/Euler_2.cpp:1455-1456).
//
// A C++ host (north_star: "host code stays C++") gets the whole exchange without Python and without staging through PCIe:
//   begin: ONE pack launch for all neighbours (k_halo_segments) on the context's stream -> event -> the TRANSPORT on a separate
//          communication stream (so interior work enqueued between begin and end overlaps the exchange);
//   end  : the context's stream waits for the transport's event -> unpack (INSERT, or ADD in rank-ordered ranges so that sums
//          whose targets repeat between neighbours are formed in a fixed order).
// Transports: RCCL (grouped ncclSend/ncclRecv over xGMI; librccl is dlopen'ed on first use -- the process's already-loaded copy when
// there is one --, the communicator is the host's), a host callback (GPU-aware MPI, torch.distributed, a test double), loop-back
// (a plan whose only neighbour is the rank itself: periodic single-rank layouts and tests), or -- round 6 -- ONE-SIDED ("peer"):
//   the receive buffers are exported with hipIpcGetMemHandle and opened by the neighbour ranks when the plan is connected; begin's pack
//   kernel writes every message STRAIGHT INTO the neighbour's receive buffer (over xGMI between GPUs of a node) and its last block
//   publishes the exchange's sequence number into the neighbour's arrival flags (system-scope release); end's unpack kernel first waits for
//   the flags of all neighbours (system-scope acquire, bounded spin).  TWO kernels per exchange, no library call on the
//   critical path, no communication stream, and nothing but kernel nodes -- the exchange can be recorded in a hipGraph with the solver
//   around it, which a kB-sized message inside a 20 us step needs to have any chance against an RCCL launch.  Receive buffers are double
//   (exchange parity): a neighbour may already write exchange k + 1 while this rank still unpacks exchange k; it cannot reach k + 2 before
//   it has seen this rank's message k + 1, which this rank packs after that unpack (stream order).
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    int transport = 0;                                       // 0 none, 1 callback, 2 loop-back, 3 RCCL, 4 one-sided (peer)
    // ---- one-sided transport: [2][nr*max_nlev] doubles of receive buffer + MIMSEM_HALO_MAX_SEGMENTS arrival flags + 1 error word, ONE allocation (one IPC handle)
    double* d_peer = nullptr; size_t peer_half = 0;          // doubles per half
    unsigned long long* d_flags = nullptr;                   // my arrival flags [MAX_SEGMENTS] + error word, inside d_peer
    void* peer_base[MIMSEM_HALO_MAX_SEGMENTS] = {};          // neighbours' allocations as opened here (hipIpcOpenMemHandle)
    size_t peer_half_of[MIMSEM_HALO_MAX_SEGMENTS] = {};      // their doubles per half
    long long peer_off[MIMSEM_HALO_MAX_SEGMENTS] = {};       // where MY message starts in neighbour i's buffer (slots per level)
    int peer_slot[MIMSEM_HALO_MAX_SEGMENTS] = {};            // which of neighbour i's flags is mine
    int my_rank = -1;
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

// ---- one-sided transport: kernels ------------------------------------------------------------------------------------------------------
struct PeerSegs { int nseg; int off[MIMSEM_HALO_MAX_SEGMENTS + 1]; double* dst[MIMSEM_HALO_MAX_SEGMENTS]; long long half[MIMSEM_HALO_MAX_SEGMENTS]; };
// The exchange counter lives in DEVICE memory (word SEQ of the flag block) and is advanced by the publish kernel: nothing of an exchange is
// baked into kernel arguments, so a recorded exchange replays correctly (exchange k of a replay uses parity and sequence number k).
constexpr int PEER_ERR = MIMSEM_HALO_MAX_SEGMENTS, PEER_SEQ = MIMSEM_HALO_MAX_SEGMENTS + 1, PEER_CNT = MIMSEM_HALO_MAX_SEGMENTS + 2, PEER_WORDS = MIMSEM_HALO_MAX_SEGMENTS + 3;
struct PeerFlags { int n; unsigned long long* flag[MIMSEM_HALO_MAX_SEGMENTS]; };
// begin = ONE kernel: entry (level, slot j of segment s) of v goes to neighbour s's receive buffer (the half of this exchange's parity, [level][slot]
// inside its message); the block that finishes LAST (a device-scope counter behind a system-scope fence: every block's stores are visible before
// the counter says so) publishes the exchange's sequence number into every neighbour's arrival flags with a system-scope release and advances
// the counter of exchanges.  total == 0 (nothing to send, neighbours all the same): one block that only publishes.
__global__ __launch_bounds__(256) void k_halo_pack_peer(PeerSegs sg, PeerFlags pf, int nlev, const int* __restrict__ idx, const double* __restrict__ v, long long vs,
                                                        unsigned long long* myflags) {
    const int total = sg.off[sg.nseg];
    const long long t = (long long)blockIdx.x*256 + threadIdx.x;
    const unsigned long long seq = myflags[PEER_SEQ] + 1;             // (advanced by the last block below, after every block has read it)
    if (t < (long long)total*nlev) {
        const int gi = (int)(t%total), lev = (int)(t/total);
        int s = 0;
        while (gi >= sg.off[s + 1]) s++;
        const int cnt = sg.off[s + 1] - sg.off[s];
        sg.dst[s][(size_t)(seq & 1)*sg.half[s] + (size_t)lev*cnt + (gi - sg.off[s])] = v[(size_t)lev*vs + idx[gi]];
    }
    __threadfence_system();
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) last = atomicAdd((unsigned long long*)(myflags + PEER_CNT), 1ULL) == (unsigned long long)gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    if ((int)threadIdx.x < pf.n) {
        __threadfence_system();
        __hip_atomic_store(pf.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (threadIdx.x == 0) { myflags[PEER_CNT] = 0; myflags[PEER_SEQ] = seq; }
}
// end = the unpack kernel itself waits: the first lanes of every block poll the arrival flags of ALL neighbours (system-scope acquire, s_sleep
// between, bounded: after ~2 s the error word is set and the block goes on -- every wave reaches an exit whatever the neighbours do), then the
// block unpacks from the half of the exchange's parity (k_halo_segments with the buffer chosen on the device): mode 1 insert, 2 add
struct PeerRecv { int nseg; int off[MIMSEM_HALO_MAX_SEGMENTS + 1]; };
__global__ __launch_bounds__(256) void k_halo_unpack_peer(PeerRecv sg, int s_begin, int s_end, int nlev, int mode, const int* __restrict__ idx,
                                                          const double* __restrict__ base, long long half, unsigned long long* myflags,
                                                          double* __restrict__ v, long long vs) {
    const unsigned long long seq = myflags[PEER_SEQ];
    if ((int)threadIdx.x < sg.nseg) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(myflags + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 200000000LL) { __hip_atomic_store(myflags + PEER_ERR, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }      // (100 MHz constant clock: 2 s)
        }
    }
    __syncthreads();
    const int g0 = sg.off[s_begin], total = sg.off[s_end] - g0;
    const long long t = (long long)blockIdx.x*256 + threadIdx.x;
    if (t >= (long long)total*nlev) return;
    const double* buf = base + (size_t)(seq & 1)*half;
    const int gi = g0 + (int)(t%total), lev = (int)(t/total);
    int s = s_begin;
    while (gi >= sg.off[s + 1]) s++;
    const int cnt = sg.off[s + 1] - sg.off[s];
    const double b = __builtin_nontemporal_load(buf + (size_t)sg.off[s]*nlev + (size_t)lev*cnt + (gi - sg.off[s]));
    double* o = v + (size_t)lev*vs + idx[gi];
    if (mode == 1) *o = b; else *o += b;
}
struct PeerBlob {                                                      // what a rank tells its neighbours about a plan (fits MIMSEM_HALO_PEER_BLOB)
    unsigned magic; int rank, nneigh, max_nlev; long long half;        // doubles per half of the receive allocation
    int ranks[MIMSEM_HALO_MAX_SEGMENTS]; int recv_off[MIMSEM_HALO_MAX_SEGMENTS + 1];
    hipIpcMemHandle_t mem;
};
static_assert(sizeof(PeerBlob) <= MIMSEM_HALO_PEER_BLOB, "PeerBlob must fit the blob the header promises");
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
    for (int i = 0; i < MIMSEM_HALO_MAX_SEGMENTS; i++) if (h->peer_base[i]) (void)hipIpcCloseMemHandle(h->peer_base[i]);
    void* ptrs[] = {h->d_send_idx, h->d_recv_idx, h->d_send, h->d_recv, h->d_peer};
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

// ---- one-sided transport: set-up -------------------------------------------------------------------------------------------------------
int mimsem_halo_peer_export(mimsem_halo* h, int my_rank, void* blob) {
    if (!h || !blob || my_rank < 0 || h->in_flight) return MIMSEM_ERR_ARG;
    mimsem_ctx* c = h->c;
    MIMSEM_HIP_TRY(hipSetDevice(c->device));
    if (!h->d_peer) {
        const size_t nr = (size_t)std::max(h->nneigh ? h->recv_off[h->nneigh] : 0, 1);
        h->peer_half = (nr*h->max_nlev + 1) & ~(size_t)1;
        const size_t bytes = 2*h->peer_half*sizeof(double) + PEER_WORDS*sizeof(unsigned long long);
        // UNCACHED device memory (MTYPE UC: not held in this GPU's L2): what a neighbour GPU writes over xGMI -- the message and the arrival flag --
        // must be seen by loads of a kernel that is already running here; an ordinary (coarse-grained) allocation is coherent with other agents at
        // kernel boundaries only, and a poll or a re-read of a buffer half could be served from a stale L2 line for ever.  (Two processes on ONE
        // GPU share its L2 and would not notice the difference: the property matters between GPUs.)  Ordinary memory is the fallback if the
        // runtime refuses the flag.
        bool uncached = true;
        if (hipExtMallocWithFlags((void**)&h->d_peer, bytes, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            uncached = false;
            MIMSEM_HIP_TRY(hipMalloc((void**)&h->d_peer, bytes));
        }
        if (getenv("MIMSEM_VERBOSE")) fprintf(stderr, "[mimsem] halo plan %p: one-sided receive buffer %zu bytes, %s device memory\n", (void*)h, bytes, uncached ? "UNCACHED" : "ordinary (hipExtMallocWithFlags refused)");
        MIMSEM_HIP_TRY(hipMemset(h->d_peer, 0, bytes));
        h->d_flags = (unsigned long long*)(h->d_peer + 2*h->peer_half);
    }
    PeerBlob b{};
    b.magic = 0x4d48504bu; b.rank = my_rank; b.nneigh = h->nneigh; b.max_nlev = h->max_nlev; b.half = (long long)h->peer_half;
    for (int i = 0; i < h->nneigh; i++) b.ranks[i] = h->ranks[i];
    for (int i = 0; i <= h->nneigh; i++) b.recv_off[i] = h->recv_off[i];
    MIMSEM_HIP_TRY(hipIpcGetMemHandle(&b.mem, h->d_peer));
    std::memset(blob, 0, MIMSEM_HALO_PEER_BLOB);
    std::memcpy(blob, &b, sizeof b);
    h->my_rank = my_rank;
    return MIMSEM_OK;
}
int mimsem_halo_set_peer(mimsem_halo* h, int my_rank, const void* neighbour_blobs) {
    if (!h || h->in_flight || my_rank < 0 || (h->nneigh && !neighbour_blobs)) return MIMSEM_ERR_ARG;
    if (!h->d_peer || h->my_rank != my_rank) return MIMSEM_ERR_STATE;      // export first (the neighbours need this rank's handle too)
    mimsem_ctx* c = h->c;
    MIMSEM_HIP_TRY(hipSetDevice(c->device));
    for (int i = 0; i < h->nneigh; i++) {
        PeerBlob b;
        std::memcpy(&b, (const char*)neighbour_blobs + (size_t)i*MIMSEM_HALO_PEER_BLOB, sizeof b);
        if (b.magic != 0x4d48504bu || b.rank != h->ranks[i] || b.nneigh < 0 || b.nneigh > MIMSEM_HALO_MAX_SEGMENTS || b.max_nlev != h->max_nlev) return MIMSEM_ERR_ARG;
        int j = -1;
        for (int k = 0; k < b.nneigh; k++) if (b.ranks[k] == my_rank) j = k;
        if (j < 0) return MIMSEM_ERR_ARG;                                // the neighbour does not list this rank
        if (b.recv_off[j + 1] - b.recv_off[j] != h->send_off[i + 1] - h->send_off[i]) return MIMSEM_ERR_ARG;      // what it expects is not what this rank sends
        if (h->peer_base[i]) { (void)hipIpcCloseMemHandle(h->peer_base[i]); h->peer_base[i] = nullptr; }
        if (b.rank == my_rank) h->peer_base[i] = nullptr;                // (a rank listed as its own neighbour: its own buffer, no handle to open)
        else MIMSEM_HIP_TRY(hipIpcOpenMemHandle(&h->peer_base[i], b.mem, hipIpcMemLazyEnablePeerAccess));
        h->peer_half_of[i] = (size_t)b.half; h->peer_off[i] = b.recv_off[j]; h->peer_slot[i] = j;
    }
    h->transport = 4;
    return MIMSEM_OK;
}

int mimsem_halo_begin(mimsem_halo* h, int mode, int nlev, double* v, long long vs) {
    if (!h || !v || nlev < 1 || nlev > h->max_nlev || (mode != MIMSEM_HALO_INSERT && mode != MIMSEM_HALO_ADD)) return MIMSEM_ERR_ARG;
    if (h->in_flight || h->transport == 0) return MIMSEM_ERR_STATE;
    mimsem_ctx* c = h->c;
    int rc;
    if (h->transport == 4) {
        // one-sided: pack straight into the neighbours' receive buffers (the half of this exchange's parity), then publish the sequence number
        if (h->nneigh) {
            PeerSegs sg; sg.nseg = h->nneigh;
            PeerFlags pf; pf.n = h->nneigh;
            for (int i = 0; i <= h->nneigh; i++) sg.off[i] = h->send_off[i];
            for (int i = 0; i < h->nneigh; i++) {
                double* base = h->peer_base[i] ? (double*)h->peer_base[i] : h->d_peer;
                sg.dst[i] = base + (size_t)h->peer_off[i]*nlev; sg.half[i] = (long long)h->peer_half_of[i];
                pf.flag[i] = (unsigned long long*)(base + 2*h->peer_half_of[i]) + h->peer_slot[i];
            }
            const long long total = (long long)h->send_off[h->nneigh]*nlev;
            hipLaunchKernelGGL(k_halo_pack_peer, dim3((unsigned)std::max<long long>(1, (total + 255)/256)), dim3(256), 0, c->stream, sg, pf, nlev, h->d_send_idx, v, vs, h->d_flags);
        }
        MIMSEM_HIP_TRY(hipGetLastError());
        h->in_flight = true; h->mode = mode; h->nlev = nlev; h->v = v; h->vs = vs;
        return MIMSEM_OK;
    }
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
    if (h->transport == 4) {
        if (!h->nneigh) return MIMSEM_OK;
        if (h->recv_off[h->nneigh]) {
            PeerRecv sg; sg.nseg = h->nneigh;
            for (int i = 0; i <= h->nneigh; i++) sg.off[i] = h->recv_off[i];
            auto unpack = [&](int b, int e, int md) {
                const long long total = (long long)(h->recv_off[e] - h->recv_off[b])*h->nlev;
                if (total > 0) hipLaunchKernelGGL(k_halo_unpack_peer, dim3((unsigned)((total + 255)/256)), dim3(256), 0, c->stream, sg, b, e, h->nlev, md,
                                                  h->d_recv_idx, h->d_peer, (long long)h->peer_half, h->d_flags, h->v, h->vs);
            };
            if (h->mode == MIMSEM_HALO_INSERT) unpack(0, h->nneigh, 1);
            else for (auto& r : h->add_ranges) unpack(r.first, r.second, 2);      // fixed order => reproducible sums
        }
        MIMSEM_HIP_TRY(hipGetLastError());
        return MIMSEM_OK;
    }
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


// the error word of the one-sided transport: non-zero = the sequence number of an exchange whose wait gave up after ~2 s (a neighbour that
// never published); synchronises the context's stream
int mimsem_halo_peer_status(mimsem_halo* h, unsigned long long* timed_out_seq) {
    if (!h || !timed_out_seq) return MIMSEM_ERR_ARG;
    *timed_out_seq = 0;
    if (h->transport != 4 || !h->d_flags) return MIMSEM_OK;
    MIMSEM_HIP_TRY(hipStreamSynchronize(h->c->stream));
    MIMSEM_HIP_TRY(hipMemcpy(timed_out_seq, h->d_flags + PEER_ERR, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return MIMSEM_OK;
}

}  // extern "C"
