// tests/cpp/thread_ranks.hpp -- several RANKS as THREADS of one process on the one GPU (test_sw_sharded.cpp, test_horiz_sharded.cpp): each thread
// holds its own Mesh (context), Shard and solver object, as every MPI rank of a real host would; the halo transport and the all-reduce are host
// callbacks that meet at a barrier and hand the messages over through host memory (a plain-MPI host would call MPI_Sendrecv / MPI_Allreduce
// there; RCCL refuses two ranks on one device).  Every rank of such a test must neighbour every other (all ranks meet in every exchange).
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/mimsem_hip.h"

namespace thread_ranks {
struct Barrier {                                     // (C++17: no std::barrier)
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m_);
        const long gen = gen_;
        if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen_ != gen; });
    }
    std::mutex m_; std::condition_variable cv_; int n_, count_ = 0; long gen_ = 0;
};
struct World {
    int size; Barrier bar;
    std::vector<std::vector<std::vector<double>>> box;          // box[from][to]: the message in flight
    std::vector<std::vector<double>> red;                        // all-reduce: every rank's contribution
    explicit World(int n) : size(n), bar(n), box(n, std::vector<std::vector<double>>(n)), red(n) {}
};
struct RankCtx { World* w; int rank; mimsem_ctx* ctx; long transports = 0, reductions = 0; };

// mimsem_halo_transport_fn: messages through host memory, all ranks meet twice (every rank of this test neighbours every other)
inline int transport(void* user, const double* send, const long long* so, double* recv, const long long* ro, int nn, const int* ranks, void*) {
    RankCtx* r = (RankCtx*)user; World* w = r->w;
    r->transports++;
    for (int i = 0; i < nn; i++) {
        auto& b = w->box[r->rank][ranks[i]];
        b.resize((size_t)(so[i + 1] - so[i]));
        if (!b.empty() && mimsem_memcpy_d2h(r->ctx, b.data(), send + so[i], (long long)b.size()*8) != MIMSEM_OK) return 1;      // (ordered after the pack: same stream)
    }
    w->bar.wait();
    for (int i = 0; i < nn; i++) {
        const auto& b = w->box[ranks[i]][r->rank];
        if ((long long)b.size() != ro[i + 1] - ro[i]) return 1;
        if (!b.empty() && mimsem_memcpy_h2d(r->ctx, recv + ro[i], b.data(), (long long)b.size()*8) != MIMSEM_OK) return 1;      // (the unpack follows on the same stream)
    }
    w->bar.wait();
    return 0;
}
inline int allreduce(void* user, double* v, int n) {
    RankCtx* r = (RankCtx*)user; World* w = r->w;
    r->reductions++;
    w->red[r->rank].assign(v, v + n);
    w->bar.wait();
    for (int i = 0; i < n; i++) { double s = 0.0; for (int k = 0; k < w->size; k++) s += w->red[k][i]; v[i] = s; }      // rank order on every rank: the same bits
    w->bar.wait();
    return 0;
}
}  // namespace thread_ranks
