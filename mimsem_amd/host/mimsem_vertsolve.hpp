// mimsem_vertsolve.hpp -- the vertical implicit solve (the caller of rows C5-C8) driven from C++ over the C ABI: the Newton loop of
// VertSolve::solve_schur_eta (eul/VertSolve.cpp:1721-1973) for EVERY column at once, on the fused entry points of the library --
// per iteration mimsem_column_newton_residual (assemble_residual_ec with diagnose_F_z / diagnose_Phi_z, the EOS and entropy residuals:
// :237-286, :432-502, :1806-1851), mimsem_column_solve_schur_eta (:677-823: the linear solve), mimsem_column_newton_update (:1858-1912) and
// mimsem_column_diag_theta_blend (diagTheta2 / diagTheta_L2 :289-352 with the half-time blend) -- and the max-norms of VertSolve::MaxNorm
// (:228: mimsem_column_max_norms) with the reference's stopping test.  All state in the "vertical" layout of L2Vecs::vz: [nEl][slots*n2e], velz on the nk-1
// interfaces, theta on nk+1, the rest on the nk levels.  Orders 1..4 (the fused entries' range).  Header-only, C++17, no HIP toolchain.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include "mimsem_shim.hpp"

namespace mimsem_host {

class VertSolveEta {
public:
    struct Norms { double exner, w, rho, eta; };
    std::vector<Norms> history;                       // |d x| / |x| (max over the columns) of the iterations of the last solve
    double k2i_z = 0.0;                               // VertSolve::k2i_z of the last iteration
    double rayleigh = 4.0/120.0;                      // RAYLEIGH, eul/VertSolve.cpp:32 (the sponge layer at the model top; 0 switches it off)
    // the horizontal transport tendencies the loop adds (:1799, :1822-1826): forcing(rho_i, rho_j, theta_l2_h, add_rho, add_rt) fills the two
    // output arrays ([nEl][nk*n2e]); empty = no horizontal wind
    std::function<void(const double*, const double*, const double*, double*, double*)> horiz_forcing;
    // Several ranks (round 6): the columns of an element live on ONE rank, so the solve needs no exchange at all -- what crosses
    // ranks is the reference's MPI_Allreduce(MAX) of the four update norms per iteration (eul/VertSolve.cpp:1915-1918: every rank must stop at
    // the same iteration) and the sum of k2i_z.  A host with more than one rank sets the two callbacks (in place, n host doubles; MPI_Allreduce
    // with MPI_MAX / MPI_SUM on MPI_COMM_WORLD); unset = one rank.
    std::function<void(double*, int)> allreduce_max, allreduce_sum;

    VertSolveEta(Mesh* m, double dt_) : mesh(m), dt(dt_) {
        nEl = m->nEl_; n2 = m->n2e; nk = m->nk_;
        nl = (size_t)nEl*nk*n2; ni = (size_t)nEl*(nk - 1)*n2; nt = (size_t)nEl*(nk + 1)*n2;
        try {
            for (double** p : {&velz_j, &velz_h, &F_w, &d_w, &k2i}) *p = mesh->device_alloc(ni);
            for (double** p : {&rho_j, &rt_j, &exner_j, &rho_h, &rt_h, &exner_h, &theta_l2_i, &theta_l2_h, &F_rho, &F_eta, &F_exner, &d_rho, &d_eta, &d_exner,
                               &th_w3, &eta, &add_rho, &add_rt, &ones}) *p = mesh->device_alloc(nl);
            for (double** p : {&theta_i, &theta_h}) *p = mesh->device_alloc(nt);
            nrm = mesh->device_alloc(8*nl); sums = mesh->device_alloc((size_t)8*nEl);
            // a row of ones: the column sums of the update's squares are row dots with it
            check(mimsem_memset(mesh->ctx, th_w3, 0, (long long)nl*8), "mimsem_memset");
            std::vector<double> one((size_t)nk*n2, 1.0);
            check(mimsem_memcpy_h2d(mesh->ctx, ones, one.data(), (long long)one.size()*8), "h2d");
        } catch (...) { release(); throw; }
    }
    ~VertSolveEta() { release(); }
    VertSolveEta(const VertSolveEta&) = delete; VertSolveEta& operator=(const VertSolveEta&) = delete;

    // VertSolve::solve_schur_eta: velz / rho / rt / exner at the old time level in, at the new one out (in place); zv from VertSolve::initGZ.
    // udwdx (nullable, [nEl][(nk-1)*n2e]): the u dw/dx term (:1809); hs_lat (nullable, [nEl][mp12]): the Held-Suarez temperature forcing.
    // Returns the number of iterations run; theta_h / theta_l2_h / exner_h (what the horizontal corrector reads) stay in the accessors below.
    int solve_schur_eta(double* velz, double* rho, double* rt, double* exner, const double* zv, int maxit = 20, double tol = 1.0e-12,
                        const double* udwdx = nullptr, const double* hs_lat = nullptr) {
        mimsem_ctx* c = mesh->ctx;
        copy(velz_j, velz, ni); copy(rho_j, rho, nl); copy(rt_j, rt, nl); copy(exner_j, exner, nl);
        check(mimsem_column_diag_theta_blend(c, rho, rt, theta_i, nullptr, theta_l2_i, nullptr, 1.0, 0.0), "diag_theta_blend");        // diagTheta2 :1766, diagTheta_L2 :1773
        copy(theta_h, theta_i, nt); copy(theta_l2_h, theta_l2_i, nl);
        copy(exner_h, exner, nl); copy(velz_h, velz, ni); copy(rho_h, rho, nl); copy(rt_h, rt, nl);
        history.clear();
        int it = 0;
        for (it = 1; it <= maxit; it++) {
            const double *a_rho = nullptr, *a_rt = nullptr;
            if (horiz_forcing) { horiz_forcing(rho, rho_j, theta_l2_h, add_rho, add_rt); a_rho = add_rho; a_rt = add_rt; }
            if (hs_lat) {                                                                                                                // :1831-1834
                double* dst = a_rt ? th_w3 : add_rt;                       // (th_w3 is free until the residual call below fills it)
                check(mimsem_column_temp_forcing_hs(c, hs_lat, exner_h, theta_h, rho_h, dst), "temp_forcing_hs");
                if (a_rt) check(mimsem_vec_combine(c, 1, (long long)nl, 1.0, th_w3, 0, 0, nullptr, 0, 1.0, add_rt, 0, add_rt, 0), "vec_combine");
                a_rt = add_rt;
            }
            check(mimsem_column_newton_residual(c, dt, rayleigh, theta_l2_h, exner_h, velz, velz_j, rho, rho_j, zv, rt, rt_j, rho_h, rt_h, exner_j,
                                                udwdx, a_rho, a_rt, F_w, F_rho, F_eta, F_exner, th_w3, eta, k2i), "newton_residual");
            check(mimsem_column_solve_schur_eta(c, dt, th_w3, rho_h, eta, exner_h, F_w, F_rho, F_eta, F_exner, d_w, d_rho, d_eta, d_exner), "solve_schur_eta");   // :1855
            check(mimsem_column_newton_update(c, d_w, d_rho, d_eta, d_exner, velz, rho, rt, exner, velz_j, rho_j, rt_j, exner_j,
                                              velz_h, rho_h, rt_h, exner_h, nrm), "newton_update");
            // MaxNorm (:228): per column sqrt(sum d^2 / sum x^2), the maximum over the columns (the rank-local part of the MPI_Allreduce(MAX), :1915-1918)
            check(mimsem_column_max_norms(c, nrm, sums, sums + 4*(size_t)nEl), "column_max_norms");
            // (the theta diagnosis does not depend on the norms: launched before the host waits for them)
            check(mimsem_column_diag_theta_blend(c, rho_j, rt_j, theta_h, theta_i, theta_l2_h, theta_l2_i, 0.5, 0.5), "diag_theta_blend");  // :1896-1912
            double mx[4];
            mesh->to_host(mx, sums + 4*(size_t)nEl, 4);
            if (allreduce_max) allreduce_max(mx, 4);                                                                                    // :1915-1918
            history.push_back({mx[0], mx[1], mx[2], mx[3]});
            if (mx[0] < tol && mx[2] < tol) break;
        }
        // k2i_z = sum(F_z . VA(theta) grad Pi) / SCALE of the last iteration
        {
            double s = 0.0;
            check(mimsem_krylov_rowdot(c, 1, (long long)ni, k2i, (long long)ni, onesi(), 0, sums), "krylov_rowdot");
            mesh->to_host(&s, sums, 1);
            if (allreduce_sum) allreduce_sum(&s, 1);
            k2i_z = s/1.0e8;
        }
        copy(velz, velz_j, ni); copy(rho, rho_j, nl); copy(rt, rt_j, nl); copy(exner, exner_j, nl);
        return std::min(it, maxit);
    }
    const double* theta_half() const { return theta_h; }          // [nEl][(nk+1)*n2e]
    const double* theta_l2_half() const { return theta_l2_h; }    // [nEl][nk*n2e]
    const double* exner_half() const { return exner_h; }

private:
    Mesh* mesh; double dt;
    int nEl = 0, n2 = 0, nk = 0; size_t nl = 0, ni = 0, nt = 0;
    double *velz_j = nullptr, *velz_h = nullptr, *F_w = nullptr, *d_w = nullptr, *k2i = nullptr;
    double *rho_j = nullptr, *rt_j = nullptr, *exner_j = nullptr, *rho_h = nullptr, *rt_h = nullptr, *exner_h = nullptr, *theta_l2_i = nullptr, *theta_l2_h = nullptr,
           *F_rho = nullptr, *F_eta = nullptr, *F_exner = nullptr, *d_rho = nullptr, *d_eta = nullptr, *d_exner = nullptr, *th_w3 = nullptr, *eta = nullptr,
           *add_rho = nullptr, *add_rt = nullptr, *ones = nullptr;
    double *theta_i = nullptr, *theta_h = nullptr, *nrm = nullptr, *sums = nullptr, *ones_i = nullptr;
    void copy(double* dst, const double* src, size_t n) { check(mimsem_vec_combine(mesh->ctx, 1, (long long)n, 1.0, src, 0, 0, nullptr, 0, 0.0, nullptr, 0, dst, 0), "vec_combine"); }
    // a vector of ones as long as an interface field (for the sum of k2i): built on first use
    const double* onesi() {
        if (!ones_i) {
            ones_i = mesh->device_alloc(ni);
            std::vector<double> one(ni, 1.0);
            check(mimsem_memcpy_h2d(mesh->ctx, ones_i, one.data(), (long long)ni*8), "h2d");
        }
        return ones_i;
    }
    void release() {
        for (double** p : {&velz_j, &velz_h, &F_w, &d_w, &k2i, &rho_j, &rt_j, &exner_j, &rho_h, &rt_h, &exner_h, &theta_l2_i, &theta_l2_h, &F_rho, &F_eta, &F_exner,
                           &d_rho, &d_eta, &d_exner, &th_w3, &eta, &add_rho, &add_rt, &ones, &theta_i, &theta_h, &nrm, &sums, &ones_i}) { if (*p) mimsem_free(*p); *p = nullptr; }
    }
};

}  // namespace mimsem_host
