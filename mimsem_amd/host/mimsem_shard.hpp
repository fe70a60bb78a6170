// mimsem_shard.hpp -- a rank's share of the exchanges and reductions when the mesh is dealt to several ranks: the halo plans that
// replace VecScatter on gtol_0 / gtol_1 (eul/Topo.cpp:145-155), the ownership weights of the inner products, the host's all-reduce, and an
// Arnoldi process with all-reduced dots for the spectral regions of the fixed-length solves.  Used by src::SWEqn (mimsem_sweqn.hpp) and
// HorizSolve (mimsem_horizsolve.hpp).  Header-only, C++17, no HIP toolchain needed.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <random>
#include "mimsem_shim.hpp"

namespace mimsem_host {

// A rank's share of the exchanges and reductions when the mesh is dealt to several ranks: what replaces VecScatter on gtol_0 /
// gtol_1 (eul/Topo.cpp:145-155) and MPI_Allreduce in the reference's distributed solves.  Slot lists as for VecScatterHalo: per neighbour
// rank the local slots this rank holds as GHOSTS of DoFs that rank owns, and the local slots this rank OWNS that the neighbour ghosts
// ("mirrors"), both in the order of the shared global numbering.
//   edges (1-forms): an edge borders at most two elements, so a shared edge has exactly two sharers: ONE symmetric exchange completes a
//     vector of element-local partial sums -- each sharer sends its partial sum to the other and both add (a + b = b + a bit for bit);
//   nodes (0-forms): patch corners have more than two sharers: REVERSE/ADD to the owner, then FORWARD/INSERT back.
// own0 / own1: 1 for the DoFs this rank owns, 0 for its ghosts -- inner products count every global DoF once.
class Shard {
public:
    using allreduce_fn = int (*)(void* user, double* v, int n);        // in-place sum over the ranks of n host doubles; 0 = ok
    Shard(Mesh* m, const std::vector<int>& ranks, const std::vector<int>& ghost1, const std::vector<int>& ghost1_off, const std::vector<int>& mirror1,
          const std::vector<int>& mirror1_off, const std::vector<int>& ghost0, const std::vector<int>& ghost0_off, const std::vector<int>& mirror0,
          const std::vector<int>& mirror0_off, const std::vector<double>& own0_host, const std::vector<double>& own1_host, allreduce_fn ar, void* ar_user)
        : mesh(m), nodes(m, 0, ranks, ghost0, ghost0_off, mirror0, mirror0_off), reduce(ar), user(ar_user) {
        const int nn = (int)ranks.size();
        std::vector<int> pidx, poff(1, 0);
        for (int i = 0; i < nn; i++) {                                  // every slot shared with neighbour i, in slot (= global) order on both sides
            std::vector<int> s(ghost1.begin() + ghost1_off[i], ghost1.begin() + ghost1_off[i + 1]);
            s.insert(s.end(), mirror1.begin() + mirror1_off[i], mirror1.begin() + mirror1_off[i + 1]);
            std::sort(s.begin(), s.end());
            pidx.insert(pidx.end(), s.begin(), s.end()); poff.push_back((int)pidx.size());
        }
        check(mimsem_halo_create(m->ctx, nn, ranks.data(), pidx.data(), poff.data(), pidx.data(), poff.data(), m->n1, m->nk_, &pair), "mimsem_halo_create(pair)");
        std::vector<int> shared(pidx); std::sort(shared.begin(), shared.end()); shared.erase(std::unique(shared.begin(), shared.end()), shared.end());
        // the shared edges: their element groups go first in the operators' plans, and the block preconditioners weight them by their GLOBAL multiplicity
        check(mimsem_ctx_set_halo_slots(m->ctx, 1, shared.data(), (int)shared.size()), "mimsem_ctx_set_halo_slots");
        if ((int)own0_host.size() != m->n0 || (int)own1_host.size() != m->n1) throw std::runtime_error("Shard: ownership weights of the wrong length");
        own0 = m->to_device(own0_host.data(), own0_host.size());
        std::vector<double> ox(own1_host); ox.resize((size_t)m->n1 + m->n2, 1.0);          // packed [u | h]: 2-forms are never shared
        ownx = m->to_device(ox.data(), ox.size()); own1 = ownx;
    }
    ~Shard() { mimsem_halo_destroy(pair); if (own0) mimsem_free(own0); if (ownx) mimsem_free(ownx); }
    Shard(const Shard&) = delete; Shard& operator=(const Shard&) = delete;
    void use_transport(mimsem_halo_transport_fn fn, void* u) { nodes.use_transport(fn, u); check(mimsem_halo_set_transport(pair, fn, u), "set_transport"); }
    void use_rccl(void* nccl_comm) { nodes.use_rccl(nccl_comm); check(mimsem_halo_set_rccl(pair, nccl_comm), "set_rccl"); }
    // complete a vector of element-local partial sums: nlev level rows, n1 (n0) doubles apart
    void complete1(double* v, int nlev = 1) { exchanges++; check(mimsem_halo_begin(pair, MIMSEM_HALO_ADD, nlev, v, mesh->n1), "halo_begin"); check(mimsem_halo_end(pair), "halo_end"); }
    void complete0(double* v, int nlev = 1) { exchanges += 2; nodes.reverse_add(v, nlev, mesh->n0); nodes.forward_insert(v, nlev, mesh->n0); }
    void allreduce(double* v, int n) { allreduces++; if (reduce && reduce(user, v, n) != 0) throw std::runtime_error("Shard: the host's all-reduce failed"); }
    // <a, b> over the GLOBAL vector: ownership-weighted local part (tmp: n doubles of device scratch, out: one device double), then all-reduced
    double dot(const double* wgt, long long n, const double* a, const double* b, double* tmp, double* out) {
        check(mimsem_vec_combine(mesh->ctx, 1, n, 1.0, a, n, 1, wgt, n, 0.0, nullptr, 0, tmp, n), "mimsem_vec_combine");
        check(mimsem_krylov_rowdot(mesh->ctx, 1, n, tmp, n, b, n, out), "mimsem_krylov_rowdot");
        double v = 0.0;
        mesh->to_host(&v, out, 1);
        allreduce(&v, 1);
        return v;
    }
    // Ritz values of the operator `body(v, w)` (w = B v on COMPLETED vectors of n entries) from m Arnoldi steps: the region of the spectrum a
    // fixed-length Chebyshev iteration is built on.  Set-up path: host round trips and all-reduces per step, once per dt.
    // complete(v): makes the shared entries of a start vector agree on all sharers.
    void ritz(long long n, int m, const double* wgt, const std::function<void(const double*, double*)>& body, const std::function<void(double*)>& complete,
              double* re_min, double* re_max, double* im_max, unsigned seed = 1) {
        mimsem_ctx* c = mesh->ctx;
        double *V = mesh->device_alloc((size_t)(m + 1)*n), *w = mesh->device_alloc(n), *tmp = mesh->device_alloc(n), *h = mesh->device_alloc(m + 4);
        std::vector<double> H((size_t)(m + 1)*m, 0.0), hh(m + 2), start(n);
        try {
            std::mt19937_64 gen(seed); std::normal_distribution<double> nd;
            for (double& x : start) x = nd(gen);
            check(mimsem_memcpy_h2d(c, w, start.data(), n*8), "h2d");
            check(mimsem_vec_combine(c, 1, n, 1.0, w, n, 1, wgt, n, 0.0, nullptr, 0, w, n), "mimsem_vec_combine");       // the owner's value ...
            complete(w);                                                                                            // ... on every sharer
            double nrm = std::sqrt(dot(wgt, n, w, w, tmp, h + m + 2));
            check(mimsem_vec_combine(c, 1, n, 1.0/nrm, w, n, 0, nullptr, 0, 0.0, nullptr, 0, V, n), "mimsem_vec_combine");
            int kk = m;
            for (int j = 0; j < m; j++) {
                body(V + (size_t)j*n, w);
                for (int pass = 0; pass < 2; pass++) {                  // classical Gram-Schmidt, twice
                    check(mimsem_vec_combine(c, 1, n, 1.0, w, n, 1, wgt, n, 0.0, nullptr, 0, tmp, n), "mimsem_vec_combine");
                    check(mimsem_krylov_mdot(c, j + 1, n, V, n, tmp, h), "mimsem_krylov_mdot");
                    mesh->to_host(hh.data(), h, j + 1);
                    allreduce(hh.data(), j + 1);
                    for (int i = 0; i <= j; i++) H[(size_t)i*m + j] += hh[i];
                    check(mimsem_memcpy_h2d(c, h, hh.data(), (j + 1)*8), "h2d");
                    check(mimsem_krylov_maxpy(c, j + 1, n, V, n, h, -1.0, w), "mimsem_krylov_maxpy");
                }
                nrm = std::sqrt(dot(wgt, n, w, w, tmp, h + m + 2));
                H[(size_t)(j + 1)*m + j] = nrm;
                if (!(nrm == nrm)) throw std::runtime_error("Shard::ritz: NaN in the Arnoldi process");
                if (nrm <= 1.0e-14*std::fabs(H[0])) { kk = j + 1; break; }
                check(mimsem_vec_combine(c, 1, n, 1.0/nrm, w, n, 0, nullptr, 0, 0.0, nullptr, 0, V + (size_t)(j + 1)*n, n), "mimsem_vec_combine");
            }
            std::vector<double> a((size_t)kk*kk), wr(kk), wi(kk);
            for (int i = 0; i < kk; i++) for (int j = 0; j < kk; j++) a[(size_t)i*kk + j] = H[(size_t)i*m + j];
            check(mimsem_hessenberg_eigenvalues(kk, a.data(), wr.data(), wi.data()), "mimsem_hessenberg_eigenvalues");
            double lo = wr[0], hi = wr[0], im = 0.0;
            for (int i = 0; i < kk; i++) { lo = std::min(lo, wr[i]); hi = std::max(hi, wr[i]); im = std::max(im, std::fabs(wi[i])); }
            *re_min = lo; *re_max = hi; *im_max = im;
        } catch (...) { mimsem_free(V); mimsem_free(w); mimsem_free(tmp); mimsem_free(h); throw; }
        mimsem_free(V); mimsem_free(w); mimsem_free(tmp); mimsem_free(h);
    }
    Mesh* mesh; VecScatterHalo nodes; mimsem_halo* pair = nullptr;
    double *own0 = nullptr, *own1 = nullptr, *ownx = nullptr;
    long exchanges = 0, allreduces = 0;                                 // counters (tests: no all-reduce inside a solve)
private:
    allreduce_fn reduce; void* user;
};

}  // namespace mimsem_host
