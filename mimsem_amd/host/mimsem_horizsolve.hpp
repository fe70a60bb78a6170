// mimsem_horizsolve.hpp -- the right-hand sides of the horizontal dynamics (row N2) driven from C++ over the C ABI: the counterpart of the
// reference's HorizSolve (eul/HorizSolve.cpp: grad :208-228, curl :233-254, laplacian :256-283, diagnose_fluxes :285-327, advection_rhs_ec
// :380-417, diagnose_Phi :419-470, diagnose_q :472-493, momentum_rhs_ec :637-786) for a host that holds its fields in device memory.
// The reference loops `for (kk ...)` around a per-level assemble + MatMult + KSPSolve; here EVERY LEVEL goes through each operator in one
// call (level rows `n` doubles apart), and the ksp1 solves of all levels are ONE batched CG (mimsem_ksp_*: one block per element as in
// PCBJACOBI, the thickness of a level as a per-(level, element) factor).
// Field layout: horizontal, one row per level -- 1-forms [nk][n1], 2-forms [nk][n2], 0-forms [nk][n0]; interface quantities (velz, dudz,
// dwdx, Fz) [nk-1][.].  Header-only, C++17, no HIP toolchain needed.
// SHARDED (round 6): with a Shard (mimsem_shard.hpp) every 0/1-form result is completed over the halo -- one exchange per operator result,
// accumulations (MIMSEM_FLAG_ACCUM) through a completed temporary --, the ksp1 solves are the fixed-length Chebyshev iteration with both
// element-local sums of a sweep completed (no inner product: no all-reduce inside a solve), their spectral interval comes from Shard::ritz,
// the check log is ownership-weighted and all-reduced ONCE by verify(), and k2i() is an all-reduced weighted sum.
#pragma once
#include <cmath>
#include <utility>
#include "mimsem_shard.hpp"

namespace mimsem_host {

class HorizSolve {
public:
    static constexpr double SCALE = 1.0e8, OMEGA = 7.29212e-5, RAD_EARTH = 6371220.0;      // eul/HorizSolve.cpp:21-25
    double del2; bool do_visc; double rtol = 1.0e-14;
    int last_its = 0;
    // The ksp1 solves as a Chebyshev semi-iteration of FIXED length on the fused block sweep (mimsem_block_chebyshev_sweep): the spectrum of
    // P M1 belongs to the mesh and its layer thicknesses, so its interval is estimated once (mimsem_ksp_ritz on the object PCSetUp built) and
    // the step count for `rtol` follows -- no inner product, no host round trip: a whole right-hand-side evaluation can be recorded in a
    // Graph.  EVERY solve is checked (round 6; the reference monitors every KSPSolve, round 5 checked the first three only and a later, rougher
    // right-hand side could have lost accuracy unseen): the first sweep's update is P b, the last sweep's the preconditioned residual it saw --
    // both norms go into a slot of a small device log with ONE two-row dot (recordable in a Graph, no host round trip); verify() reads the
    // log once -- per right-hand-side evaluation or per time step, the caller's choice, at least every MAXLOG solves -- and on a miss turns
    // the fixed-length mode off (the CG of the reference's structure from then on; the caller redoes the evaluation).
    // levels_changed() after mimsem_ctx_set_levels: PCSetUp and the interval again.  use_fixed_length(false) keeps the CG.
    static constexpr int MAXLOG = 32;
    double margin_lo = 0.90, margin_hi = 1.05;      // the safety margins in force around the Ritz interval (use_fixed_length)
    bool whole_solve = true;            // one context: a mass solve is ONE mimsem_block_chebyshev_solve call (false: cheb_steps sweep calls)
    int cheb_steps = 0; bool fixed_length = false; int solves_checked = 0, solves_missed = 0; double worst_rel = 0.0;

    // fg: the Coriolis 0-form per level (HorizSolve::coriolis :124-161), device [nk][n0]; nDofs0G: the GLOBAL node count (viscosity() :112-120)
    // shard (optional): this rank's part of the exchanges and reductions; nDofs0G must then be the GLOBAL node count
    HorizSolve(Mesh* m, const double* fg_dev, long long nDofs0G = 0, bool visc = true, Shard* shard = nullptr) : mesh(m), fg(fg_dev), sh(shard) {
        nk = m->nk_; n0 = m->n0; n1 = m->n1; n2 = m->n2; do_visc = visc;
        if (sh && nDofs0G <= 0) throw std::runtime_error("HorizSolve (sharded): the global node count is needed for the viscosity");
        const double dx = std::sqrt(4.0*M_PI*RAD_EARTH*RAD_EARTH/(double)(nDofs0G > 0 ? nDofs0G : n0));
        del2 = -std::sqrt(0.072*std::pow(dx, 3.2));
        try {
            for (double** p : {&a1, &b1, &c1, &d1, &e1, &g1, &p1, &gt1, &w1, &y1, &z1}) *p = mesh->device_alloc((size_t)nk*n1);
            if (sh) {                                         // ownership weights of all levels side by side (the rows of a batched inner product)
                std::vector<double> o1(n1), on((size_t)nk*n1);
                mesh->to_host(o1.data(), sh->own1, n1);
                for (int k = 0; k < nk; k++) std::copy(o1.begin(), o1.end(), on.begin() + (size_t)k*n1);
                own1n = mesh->to_device(on.data(), on.size());
            }
            // the two vectors of a check side by side (the second row at an even offset): one two-row dot per solve
            pair1 = mesh->device_alloc(2*even((long long)nk*n1)); upd1 = pair1; pb1 = pair1 + even((long long)nk*n1);
            chk = mesh->device_alloc(2*MAXLOG);
            check(mimsem_memset(mesh->ctx, chk, 0, 2*MAXLOG*8), "mimsem_memset");
            for (double** p : {&a2, &b2, &c2}) *p = mesh->device_alloc((size_t)nk*n2);
            for (double** p : {&m0, &a0, &b0}) *p = mesh->device_alloc((size_t)nk*n0);
            scal = mesh->device_alloc(4);
            check(mimsem_pvec(mesh->ctx, 0, nk, SCALE, nullptr, 0, m0, n0), "mimsem_pvec");                   // M0 is diagonal (collocated 0-forms)
            if (sh) sh->complete0(m0, nk);
            // ksp1 (:77-96): the 1-form mass of every level, one element block each
            check(mimsem_ksp_create(mesh->ctx, MIMSEM_KSP_CG, &ksp1), "mimsem_ksp_create");
            check(mimsem_ksp_set_operator(ksp1, MIMSEM_OP_UMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT, nullptr, 0), "mimsem_ksp_set_operator");
            check(mimsem_ksp_set_pc_bjacobi(ksp1), "mimsem_ksp_set_pc_bjacobi");
            check(mimsem_ksp_set_tolerances(ksp1, rtol, 1.0e-50, 1000, 0, 2), "mimsem_ksp_set_tolerances");
            use_fixed_length(true);
        } catch (...) { release(); throw; }                  // (a constructor that throws runs no destructor)
    }
    void use_fixed_length(bool on) {
        fixed_length = false; wanted_fixed = on;
        if (!on) return;
        // the interval from TWO Ritz estimates (25 and 40 steps): what the ends still move between them is the measure of their uncertainty
        double lo = 0.0, hi = 0.0, im = 0.0, lo25 = 0.0, hi25 = 0.0;
        for (const int steps : {25, 40}) {
        lo25 = lo; hi25 = hi;
        if (sh) {
            // (the blocks first: the sharded interval is that of the COMPLETED operator, from the host's own Arnoldi process)
            if (mimsem_ksp_get_pc_blocks(ksp1, &blocks1, &escale1, nullptr) != MIMSEM_OK) throw std::runtime_error("HorizSolve (sharded): no element blocks for this order");
            sh->ritz((long long)nk*n1, steps, own1n, [&](const double* v, double* w) {
                         check(mimsem_op_apply(mesh->ctx, MIMSEM_OP_UMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT, nullptr, 0, v, n1, y1, n1, 1.0), "UMAT"); sh->complete1(y1, nk);
                         check(mimsem_elem_blocks_apply(mesh->ctx, 1, nk, 0, blocks1, 0, escale1, mesh->nEl_, y1, n1, w, n1, 1.0), "mimsem_elem_blocks_apply"); sh->complete1(w, nk); },
                     [&](double* v) { sh->complete1(v, nk); }, &lo, &hi, &im, 1234);
        } else check(mimsem_ksp_ritz(ksp1, steps, &lo, &hi, &im), "mimsem_ksp_ritz");
        }
        if (!(lo > 0.02) || mimsem_ksp_get_pc_blocks(ksp1, &blocks1, &escale1, nullptr) != MIMSEM_OK) {
            if (sh) throw std::runtime_error("HorizSolve (sharded): the spectral interval does not admit the fixed-length solves (no CG on a shard)");
            return;
        }
        // safety margins (round 6): three times what the ends moved, at least 1 %, at most the 10 % / 5 % of round 5 -- on a smooth thickness
        // field the Ritz values are exact to 1e-4 and the wide margins cost 3 of 15 steps; every solve is still checked (verify())
        margin_lo = 1.0 - std::min(0.10, std::max(0.01, 3.0*std::fabs(lo - lo25)/lo));
        margin_hi = 1.0 + std::min(0.05, std::max(0.01, 3.0*std::fabs(hi - hi25)/hi));
        const double l1 = margin_lo*lo, l2 = margin_hi*hi, sg = (std::sqrt(l2/l1) - 1.0)/(std::sqrt(l2/l1) + 1.0), d = 0.5*(l1 + l2), c2 = 0.25*(l2 - l1)*(l2 - l1);
        cheb_steps = std::max(2, (int)std::ceil(std::log(2.0/rtol)/std::log(1.0/sg)));
        coef.clear(); flat.clear();
        double al = 1.0/d;
        coef.emplace_back(al, 0.0);
        for (int k = 1; k < cheb_steps; k++) { const double be = (k == 1 ? 0.5 : 0.25)*c2*al*al; al = 1.0/(d - be/al); coef.emplace_back(al, be); }
        slot = 0; fixed_length = true;
    }
    void shorten_for_test(int steps) { if ((int)coef.size() > steps) { coef.resize(steps); cheb_steps = steps; } }      // (tests: a solve that must miss its check)
    // after mimsem_ctx_set_levels (new layer thicknesses): the element blocks and the per-(level, element) factors of the preconditioner and
    // the spectral interval belong to the old ones
    void levels_changed() {
        check(mimsem_ksp_set_pc_bjacobi(ksp1), "mimsem_ksp_set_pc_bjacobi");
        const bool was = fixed_length || wanted_fixed;
        use_fixed_length(was);
    }
    // the checks of every fixed-length solve since the last call, in one read: true = all met 30 rtol (the residual the LAST sweep saw: one more
    // contraction lies between it and the result).  false: fixed_length is off now -- redo the evaluation (it then runs the CG).  Synchronises.
    bool verify() {
        if (!fixed_length && slot == 0) return true;
        double v[2*MAXLOG];
        mesh->to_host(v, chk, 2*MAXLOG);
        if (sh) sh->allreduce(v, 2*MAXLOG);                          // ONE all-reduce for every solve since the last call
        check(mimsem_memset(mesh->ctx, chk, 0, 2*MAXLOG*8), "mimsem_memset");
        slot = 0;
        bool ok = true;
        for (int k = 0; k < MAXLOG; k++) {
            const double r2 = v[2*k], ref2 = v[2*k + 1];
            if (r2 == 0.0 && ref2 == 0.0) continue;                  // (slot not written, or a zero right-hand side)
            const double rel = ref2 > 0.0 ? std::sqrt(r2/ref2) : 1.0e300;
            solves_checked++;
            if (rel == rel && rel > worst_rel) worst_rel = rel;
            if (!(rel <= 30.0*rtol)) { ok = false; solves_missed++; }
        }
        if (!ok && sh) throw std::runtime_error("HorizSolve (sharded): a fixed-length mass solve missed its check (no CG on a shard)");
        if (!ok) fixed_length = false;                              // the interval was too optimistic for these right-hand sides: the CG from here on
        return ok;
    }
    ~HorizSolve() { release(); }
    HorizSolve(const HorizSolve&) = delete; HorizSolve& operator=(const HorizSolve&) = delete;

    // u = M1^-1 E12 M2 phi  (:208-228)
    void grad(const double* phi, double* u) {
        ap(MIMSEM_OP_WMAT, MIMSEM_FLAG_VERT, nullptr, 0, phi, n2, a2, n2, 1.0);
        inc(2, a2, n2, g1, n1);
        solve_M1(g1, u);
    }
    // w = M0^-1 E01 M1 u (+ f)  (:233-254)
    void curl(const double* u, double* w, bool add_f = false) {
        ap(MIMSEM_OP_UMAT, MIMSEM_FLAG_VERT, nullptr, 0, u, n1, g1, n1, 1.0);
        inc(3, g1, n1, w, n0);
        comb(n0, 1.0, w, 2, m0, add_f ? 1.0 : 0.0, add_f ? fg : nullptr, w);
    }
    // del2 (grad(E21 u) + E10 curl(u))  (:256-283)
    void laplacian(const double* u, double* out) {
        inc(1, u, n1, c2, n2);
        grad(c2, out);
        curl(u, b0);
        inc(0, b0, n0, e1, n1);
        comb(n1, del2, e1, 0, nullptr, del2, out, out);
    }
    // F = M1^-1 (hu),  G = M1^-1 F(theta) F  (:285-327, theta_in_Wt = false)
    void diagnose_fluxes(const double* u1, const double* u2, const double* h1, const double* h2, const double* theta, double* F, double* G) {
        uvec_hu4(u1, u2, h1, h2, d1);
        solve_M1(d1, F);
        ap(MIMSEM_OP_UHMAT, MIMSEM_FLAG_VERT, theta, n2, F, n1, d1, n1, 1.0);
        solve_M1(d1, G);
    }
    // :380-417: dF, dG [nk][n2] (horizontal layout; the caller's HorizToVert is mimsem_l2_transpose), Fk, Gk [nk][n1]
    void advection_rhs_ec(const double* u1, const double* u2, const double* h1, const double* h2, const double* theta, double* dF, double* dG,
                          double* Fk, double* Gk) {
        diagnose_fluxes(u1, u2, h1, h2, theta, Fk, Gk);
        inc(1, Fk, n1, b2, n2);
        ap(MIMSEM_OP_WMAT, MIMSEM_FLAG_VERT, nullptr, 0, b2, n2, dF, n2, 1.0);
        inc(1, Gk, n1, c2, n2);
        ap(MIMSEM_OP_WMAT, MIMSEM_FLAG_VERT, nullptr, 0, c2, n2, dG, n2, 0.5);
        ap(MIMSEM_OP_WHMAT, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, theta, n2, b2, n2, dG, n2, 0.5);
        grad(theta, gt1);                                                                                   // (kept: last_grad_theta())
        ap(MIMSEM_OP_WTQUMAT, MIMSEM_FLAG_ACCUM, gt1, n1, Fk, n1, dG, n2, 1.0);                             // K incl. its 0.5 factor
    }
    // grad(theta) of the last advection_rhs_ec [nk][n1]: momentum_rhs_ec of the same stage solves the same system for the same theta
    // (:403 and :659 in the reference, two KSPSolves with one answer) -- passed as its dTheta it saves one of the seven mass solves
    const double* last_grad_theta() const { return gt1; }
    // :419-470
    void diagnose_Phi(const double* u1, const double* u2, const double* velz1, const double* velz2, double* Phi) {
        ap(MIMSEM_OP_WTQUMAT, 0, u1, n1, u1, n1, Phi, n2, 1.0/3.0);
        ap(MIMSEM_OP_WTQUMAT, MIMSEM_FLAG_ACCUM, u1, n1, u2, n1, Phi, n2, 1.0/3.0);
        ap(MIMSEM_OP_WTQUMAT, MIMSEM_FLAG_ACCUM, u2, n1, u2, n1, Phi, n2, 1.0/3.0);
        // 0.5 (interface k-1) + 0.5 (interface k), the missing boundary interfaces left out (:451-459)
        check(mimsem_interface_average(mesh->ctx, nk, n2, velz1, n2, b2, n2), "mimsem_interface_average");
        check(mimsem_interface_average(mesh->ctx, nk, n2, velz2, n2, c2, n2), "mimsem_interface_average");
        ap(MIMSEM_OP_WHMAT, MIMSEM_FLAG_ACCUM, b2, n2, b2, n2, Phi, n2, 1.0/6.0);
        ap(MIMSEM_OP_WHMAT, MIMSEM_FLAG_ACCUM, b2, n2, c2, n2, Phi, n2, 1.0/6.0);
        ap(MIMSEM_OP_WHMAT, MIMSEM_FLAG_ACCUM, c2, n2, c2, n2, Phi, n2, 1.0/6.0);
    }
    // (M0h(rho)) q = E01 M1 u + M0 f; M0h is diagonal  (:472-493)
    void diagnose_q(const double* rho, const double* u, double* q) {
        ap(MIMSEM_OP_UMAT, MIMSEM_FLAG_VERT, nullptr, 0, u, n1, g1, n1, 1.0);
        inc(3, g1, n1, q, n0);
        comb(n0, 1.0, m0, 1, fg, 1.0, q, q);
        check(mimsem_pvec(mesh->ctx, 0, nk, SCALE, rho, n2, b0, n0), "mimsem_pvec");
        if (sh) sh->complete0(b0, nk);
        comb(n0, 1.0, q, 2, b0, 0.0, nullptr, q);
    }
    // :637-786 for every level at once: fu [nk][n1].  Optional: Fx (the mass flux, else diagnosed), Fz (vertical mass flux on the interfaces,
    // else the mean vertical velocity), dwdx1 / dwdx2, Fk (then k2i() is the kinetic-to-internal exchange :697-701)
    void momentum_rhs_ec(const double* theta, const double* dudz1, const double* dudz2, const double* velz1, const double* velz2, const double* Pi,
                         const double* velx1, const double* velx2, const double* rho1, const double* rho2, double* fu,
                         const double* Fx = nullptr, const double* Fz = nullptr, const double* dwdx1 = nullptr, const double* dwdx2 = nullptr,
                         const double* Fk = nullptr, const double* dTheta = nullptr) {
        mimsem_ctx* c = mesh->ctx;
        diagnose_Phi(velx1, velx2, velz1, velz2, a2);
        inc(2, a2, n2, fu, n1);
        grad(Pi, a1);                                                                                         // dPi
        if (!dTheta) { grad(theta, b1); dTheta = b1; }                                                        // dTheta
        comb(n1, 0.5, velx1, 0, nullptr, 0.5, velx2, c1);                                                     // uh
        comb(n2, 0.5, rho1, 0, nullptr, 0.5, rho2, b2);
        diagnose_q(b2, c1, a0);
        if (!Fx) { uvec_hu4(velx1, velx2, rho1, rho2, d1); solve_M1(d1, e1); Fx = e1; }
        ap(MIMSEM_OP_ROTMAT, MIMSEM_FLAG_ACCUM, a0, n0, Fx, n1, fu, n1, 1.0);
        ap(MIMSEM_OP_UHMAT, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, theta, n2, a1, n1, fu, n1, 0.5);            // pressure gradient force
        ap(MIMSEM_OP_UHMAT, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, Pi, n2, dTheta, n1, fu, n1, -0.5);
        ap(MIMSEM_OP_WHMAT, MIMSEM_FLAG_VERT, Pi, n2, theta, n2, a2, n2, 1.0);
        inc(2, a2, n2, d1, n1);                                                                               // dp
        comb(n1, 0.5, d1, 0, nullptr, 1.0, fu, fu);
        have_k2i = Fk != nullptr;
        if (Fk && sh) { combr(nk, n1, 1.0, Fk, 1, own1n, 0.0, nullptr, w1); check(mimsem_krylov_rowdot(c, 1, (long long)nk*n1, w1, (long long)nk*n1, d1, (long long)nk*n1, scal), "mimsem_krylov_rowdot"); }
        else if (Fk) check(mimsem_krylov_rowdot(c, 1, (long long)nk*n1, Fk, (long long)nk*n1, d1, (long long)nk*n1, scal), "mimsem_krylov_rowdot");
        // second vorticity term: interface i feeds levels i and i+1 (:704-746)
        if (nk > 1) {
            combr(nk - 1, n1, 0.5, dudz1, 0, nullptr, 0.5, dudz2, a1);                                        // dz
            if (dwdx1) { combr(nk - 1, n1, -0.5, dwdx1, 0, nullptr, 1.0, a1, a1); combr(nk - 1, n1, -0.5, dwdx2, 0, nullptr, 1.0, a1, a1); }
            const double* v = Fz;
            if (!v) { combr(nk - 1, n2, 0.5, velz1, 0, nullptr, 0.5, velz2, a2); v = a2; }
            apn(nk - 1, MIMSEM_OP_UTQWMAT, 0, a1, n1, v, n2, b1, n1, 1.0);                                      // UtQWmat::assemble(u1, scale): no thickness
            combr(nk - 1, n1, 0.5, b1, 0, nullptr, 1.0, fu + n1, fu + n1);
            combr(nk - 1, n1, 0.5, b1, 0, nullptr, 1.0, fu, fu);
        }
        if (do_visc) {
            laplacian(c1, a1);
            laplacian(a1, b1);
            ap(MIMSEM_OP_UMAT, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, nullptr, 0, b1, n1, fu, n1, 1.0);
        }
    }
    // horizontal kinetic-to-internal energy exchange of the last momentum_rhs_ec that was given Fk (synchronises)
    double k2i() {
        if (!have_k2i) return 0.0;
        double v = 0.0;
        mesh->to_host(&v, scal, 1);
        if (sh) sh->allreduce(&v, 1);
        return v/SCALE;
    }
    // KSPSolve(ksp1, b, x) for all levels
    void solve_M1(const double* b, double* x) {
        if (fixed_length) {
            mimsem_ctx* c = mesh->ctx;
            const long long tot = (long long)nk*n1;
            bool unsupported = false, whole = false;
            const size_t last = coef.size() - 1;
            if (!sh && whole_solve) {
                // one context: the whole solve as ONE call (the first step has no operator pass and clears nothing): the same bits as the sweeps below
                if (flat.size() != 2*coef.size()) { flat.clear(); for (const auto& ab : coef) { flat.push_back(ab.first); flat.push_back(ab.second); } }
                const int rc = mimsem_block_chebyshev_solve(c, MIMSEM_OP_UMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT, nullptr, 0, blocks1, escale1, mesh->nEl_, b, n1,
                                                            (int)coef.size(), flat.data(), x, n1, pb1, n1, upd1, n1);
                if (rc == MIMSEM_ERR_UNSUPPORTED) whole_solve = false;
                else { check(rc, "mimsem_block_chebyshev_solve"); whole = true; }
            }
            if (!whole) { check(mimsem_memset(c, x, 0, tot*8), "mimsem_memset"); check(mimsem_memset(c, p1, 0, tot*8), "mimsem_memset"); }
            for (size_t k = 0; k < coef.size() && !unsupported && !whole; k++) {
                // the update of sweep 0 (x = 0) is P b; the one of the last sweep the preconditioned residual it saw
                double* upd = k == last ? upd1 : (k == 0 ? pb1 : nullptr);
                if (sh) {
                    // sharded: z = P (b - M1 x) with both element-local sums completed over the halo; p = z + beta p; x += alpha p -- no inner product
                    check(mimsem_op_apply(c, MIMSEM_OP_UMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT, nullptr, 0, x, n1, y1, n1, 1.0), "UMAT");
                    sh->complete1(y1, nk);
                    comb(n1, -1.0, y1, 0, nullptr, 1.0, b, y1);
                    check(mimsem_elem_blocks_apply(c, 1, nk, 0, blocks1, 0, escale1, mesh->nEl_, y1, n1, z1, n1, 1.0), "mimsem_elem_blocks_apply");
                    sh->complete1(z1, nk);
                    check(mimsem_krylov_chebyshev_px(c, nk, n1, coef[k].first, coef[k].second, z1, n1, nullptr, 0, nullptr, 0, p1, n1, x, n1, upd, n1),
                          "mimsem_krylov_chebyshev_px");                  // p = z + beta p; x += alpha p; upd = z: one launch
                    continue;
                }
                const int rc = mimsem_block_chebyshev_sweep(c, MIMSEM_OP_UMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT, nullptr, 0, blocks1, escale1, mesh->nEl_, b, n1,
                                                            coef[k].first, coef[k].second, p1, n1, x, n1, upd, n1);
                if (k == 0 && rc == MIMSEM_ERR_UNSUPPORTED) unsupported = true;         // (an order the fused sweep does not cover: the CG below)
                else check(rc, "mimsem_block_chebyshev_sweep");
            }
            if (unsupported) fixed_length = false;
            else {
                last_its = cheb_steps;
                const int k = slot < MAXLOG ? slot++ : MAXLOG - 1;      // (more than MAXLOG solves between two verify() calls: the last slot is reused)
                if (sh) {                                              // this rank's ownership-weighted part of both norms
                    comb(n1, 1.0, upd1, 1, own1n, 0.0, nullptr, w1);
                    check(mimsem_krylov_rowdot(c, 1, tot, w1, tot, upd1, tot, chk + 2*k), "mimsem_krylov_rowdot");
                    comb(n1, 1.0, pb1, 1, own1n, 0.0, nullptr, w1);
                    check(mimsem_krylov_rowdot(c, 1, tot, w1, tot, pb1, tot, chk + 2*k + 1), "mimsem_krylov_rowdot");
                } else check(mimsem_krylov_rowdot(c, 2, tot, pair1, (long long)even(tot), pair1, (long long)even(tot), chk + 2*k), "mimsem_krylov_rowdot");
                return;
            }
        }
        if (sh) throw std::runtime_error("HorizSolve (sharded): the 1-form mass solve exists in the fixed-length mode only");
        check(mimsem_ksp_solve(ksp1, b, n1, x, n1), "mimsem_ksp_solve");
        double rn; int reason;
        check(mimsem_ksp_get_info(ksp1, &last_its, &rn, &reason), "mimsem_ksp_get_info");
        if (reason < 0) throw std::runtime_error("HorizSolve: the 1-form mass solve did not converge");
    }

private:
    void release() {
        mimsem_ksp_destroy(ksp1); ksp1 = nullptr;
        for (double** p : {&a1, &b1, &c1, &d1, &e1, &g1, &a2, &b2, &c2, &m0, &a0, &b0, &scal, &p1, &pair1, &chk, &gt1, &w1, &y1, &z1, &own1n}) { if (*p) mimsem_free(*p); *p = nullptr; }
        upd1 = pb1 = nullptr;
    }
    static size_t even(long long n) { return (size_t)((n + 1) & ~1LL); }
    Mesh* mesh; const double* fg; Shard* sh = nullptr; mimsem_ksp* ksp1 = nullptr;
    double *w1 = nullptr, *y1 = nullptr, *z1 = nullptr, *own1n = nullptr;
    int nk = 1, n0 = 0, n1 = 0, n2 = 0; bool have_k2i = false;
    const double *blocks1 = nullptr, *escale1 = nullptr; std::vector<std::pair<double, double>> coef; std::vector<double> flat; int slot = 0; bool wanted_fixed = true;
    double *p1 = nullptr, *upd1 = nullptr, *pb1 = nullptr, *pair1 = nullptr, *chk = nullptr, *gt1 = nullptr;
    double *a1 = nullptr, *b1 = nullptr, *c1 = nullptr, *d1 = nullptr, *e1 = nullptr, *g1 = nullptr, *a2 = nullptr, *b2 = nullptr, *c2 = nullptr,
           *m0 = nullptr, *a0 = nullptr, *b0 = nullptr, *scal = nullptr;
    static bool to_1form(int op) { return op == MIMSEM_OP_UMAT || op == MIMSEM_OP_UHMAT || op == MIMSEM_OP_ROTMAT || op == MIMSEM_OP_UTQWMAT || op == MIMSEM_OP_UTMAT || op == MIMSEM_OP_UTMAT_H; }
    // one operator over `rows` levels; sharded + 1-form result: the element-local sums are completed over the halo before anybody reads them -- an
    // accumulation (MIMSEM_FLAG_ACCUM) goes through a completed temporary (y holds complete values: partial sums must not be mixed into it)
    void apn(int rows, int op, unsigned flags, const double* f, long long fs, const double* x, long long xs, double* y, long long ys, double alpha) {
        if (sh && to_1form(op)) {
            if (flags & MIMSEM_FLAG_ACCUM) {
                check(mimsem_op_apply(mesh->ctx, op, 0, rows, SCALE, flags & ~(unsigned)MIMSEM_FLAG_ACCUM, f, fs, x, xs, w1, n1, alpha), "mimsem_op_apply");
                sh->complete1(w1, rows);
                combr(rows, n1, 1.0, w1, 0, nullptr, 1.0, y, y);
            } else {
                check(mimsem_op_apply(mesh->ctx, op, 0, rows, SCALE, flags, f, fs, x, xs, y, ys, alpha), "mimsem_op_apply");
                sh->complete1(y, rows);
            }
            return;
        }
        check(mimsem_op_apply(mesh->ctx, op, 0, rows, SCALE, flags, f, fs, x, xs, y, ys, alpha), "mimsem_op_apply");
    }
    void ap(int op, unsigned flags, const double* f, long long fs, const double* x, long long xs, double* y, long long ys, double alpha) { apn(nk, op, flags, f, fs, x, xs, y, ys, alpha); }
    void inc(int which, const double* x, long long xs, double* y, long long ys) {
        check(mimsem_incidence_apply(mesh->ctx, which, nk, x, xs, y, ys), "mimsem_incidence_apply");
        if (sh && (which == 0 || which == 2)) sh->complete1(y, nk);       // E10, E12: every edge computed by the element that owns it
        if (sh && which == 3) sh->complete0(y, nk);                        // E01
    }
    void combr(int rows, long long n, double a, const double* A, int op, const double* B, double b, const double* C, double* out) {
        check(mimsem_vec_combine(mesh->ctx, rows, n, a, A, n, op, B, n, b, C, n, out, n), "mimsem_vec_combine");
    }
    void comb(long long n, double a, const double* A, int op, const double* B, double b, const double* C, double* out) { combr(nk, n, a, A, op, B, b, C, out); }
    // the four m1->assemble_hu(level, SCALE, u, h, false, fac) calls (:300-305, :675-682)
    void uvec_hu4(const double* ua, const double* ub, const double* ha, const double* hb, double* hu) {
        if (sh) {                                             // the four LOCAL partial sums first, ONE exchange for their sum
            mimsem_ctx* c = mesh->ctx;
            check(mimsem_op_apply(c, MIMSEM_OP_UHMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT, ha, n2, ua, n1, hu, n1, 1.0/3.0), "UHMAT");
            check(mimsem_op_apply(c, MIMSEM_OP_UHMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, hb, n2, ua, n1, hu, n1, 1.0/6.0), "UHMAT");
            check(mimsem_op_apply(c, MIMSEM_OP_UHMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, ha, n2, ub, n1, hu, n1, 1.0/6.0), "UHMAT");
            check(mimsem_op_apply(c, MIMSEM_OP_UHMAT, 0, nk, SCALE, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, hb, n2, ub, n1, hu, n1, 1.0/3.0), "UHMAT");
            sh->complete1(hu, nk);
            return;
        }
        ap(MIMSEM_OP_UHMAT, MIMSEM_FLAG_VERT, ha, n2, ua, n1, hu, n1, 1.0/3.0);
        ap(MIMSEM_OP_UHMAT, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, hb, n2, ua, n1, hu, n1, 1.0/6.0);
        ap(MIMSEM_OP_UHMAT, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, ha, n2, ub, n1, hu, n1, 1.0/6.0);
        ap(MIMSEM_OP_UHMAT, MIMSEM_FLAG_VERT | MIMSEM_FLAG_ACCUM, hb, n2, ub, n1, hu, n1, 1.0/3.0);
    }
};

}  // namespace mimsem_host
