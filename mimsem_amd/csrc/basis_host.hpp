// mimsem_amd/csrc/basis_host.hpp -- host-side discretisation primitives of the product library.
// GLL rule, Lagrange nodal basis and edge (histopolant) basis tables, rows A1-A4 of the scope table
// (reference: eul/Basis.cpp:22-283, eul/ElMats.cpp:20-179).  Init-time only; results are uploaded once.
#pragma once
#include <cmath>
#include <vector>

namespace mimsem {

struct Gll {
    int n = 0;
    std::vector<double> x, w;
    // closed forms for n<=6, 15-digit literals for n=7 (eul/Basis.cpp:31-89); false if n unsupported
    bool init(int order) {
        n = order;
        x.assign(n + 1, 0.0); w.assign(n + 1, 0.0);
        auto sym = [&](int i, double xi, double wi) { x[i] = -xi; x[n - i] = +xi; w[i] = w[n - i] = wi; };
        switch (n) {
        case 1: sym(0, 1.0, 1.0); break;
        case 2: sym(0, 1.0, 1.0/3.0); x[1] = 0.0; w[1] = 4.0/3.0; break;
        case 3: sym(0, 1.0, 1.0/6.0); sym(1, std::sqrt(0.2), 5.0/6.0); break;
        case 4: sym(0, 1.0, 0.1); sym(1, std::sqrt(3.0/7.0), 49.0/90.0); x[2] = 0.0; w[2] = 64.0/90.0; break;
        case 5: { double a = 2.0*std::sqrt(7.0)/21.0;
                  sym(0, 1.0, 1.0/15.0); sym(1, std::sqrt(1.0/3.0 + a), (14.0 - std::sqrt(7.0))/30.0);
                  sym(2, std::sqrt(1.0/3.0 - a), (14.0 + std::sqrt(7.0))/30.0); break; }
        case 6: { double a = 2.0*std::sqrt(5.0/3.0)/11.0;
                  sym(0, 1.0, 1.0/21.0); sym(1, std::sqrt(5.0/11.0 + a), (124.0 - 7.0*std::sqrt(15.0))/350.0);
                  sym(2, std::sqrt(5.0/11.0 - a), (124.0 + 7.0*std::sqrt(15.0))/350.0); x[3] = 0.0; w[3] = 256.0/525.0; break; }
        case 7: sym(0, 1.0, 0.035714285714286); sym(1, 0.871740148509607, 0.210704227143506);
                sym(2, 0.591700181433142, 0.341122692483504); sym(3, 0.209299217902479, 0.412458794658704); break;
        default: return false;
        }
        double s = 0.0;
        for (double wi : w) s += wi;
        return std::fabs(s - 2.0) <= 1.0e-8;     // the reference's own self check, eul/Basis.cpp:91-97
    }
};

// l_i(x) on nodes xn (eul/Basis.cpp:180-187)
inline double lagrange(const std::vector<double>& xn, double x, int i) {
    double y = 1.0;
    for (int j = 0; j < (int)xn.size(); j++) if (j != i) y *= (x - xn[j])/(xn[i] - xn[j]);
    return y;
}
// l_i'(x) (eul/Basis.cpp:189-210)
inline double lagrange_deriv(const std::vector<double>& xn, double x, int i) {
    double acc = 0.0;
    int np1 = (int)xn.size();
    for (int j = 0; j < np1; j++) {
        if (j == i) continue;
        double a = 1.0;
        for (int k = 0; k < np1; k++) if (k != i && k != j) a *= (x - xn[k])/(xn[i] - xn[k]);
        acc += a/(xn[i] - xn[j]);
    }
    return acc;
}
// e_i(x) = -sum_{j<=i} l_j'(x) (eul/Basis.cpp:274-283)
inline double edge_fn(const std::vector<double>& xn, double x, int i) {
    double c = 0.0;
    for (int j = 0; j <= i; j++) c -= lagrange_deriv(xn, x, j);
    return c;
}

// tables at the quadrature points of GLL(m) for a basis on GLL(n)
struct BasisTables {
    int n = 0, m = 0;
    Gll quad, nodes;
    std::vector<double> L;   // [m+1][n+1]  ljxi
    std::vector<double> E;   // [m+1][n]    ejxi
    std::vector<double> P, U, V, W, Q;   // dense element tables (ElMats), row = quad point
    bool collocated = false;             // L == identity exactly (m == n)
    bool init(int order, int qorder) {
        n = order; m = qorder;
        if (!quad.init(m) || !nodes.init(n)) return false;
        int np1 = n + 1, mp1 = m + 1, mp12 = mp1*mp1;
        L.resize(mp1*np1); E.resize(mp1*n);
        for (int q = 0; q < mp1; q++) {
            for (int j = 0; j < np1; j++) L[q*np1 + j] = lagrange(nodes.x, quad.x[q], j);
            for (int j = 0; j < n; j++)   E[q*n + j] = edge_fn(nodes.x, quad.x[q], j);
        }
        collocated = (m == n);
        if (collocated) for (int q = 0; q < mp1; q++) for (int j = 0; j < np1; j++)
            if (L[q*np1 + j] != (q == j ? 1.0 : 0.0)) collocated = false;
        P.resize(mp12*np1*np1); U.resize(mp12*np1*n); V.resize(mp12*np1*n); W.resize(mp12*n*n); Q.resize(mp12);
        for (int q = 0; q < mp12; q++) {
            int qx = q % mp1, qy = q / mp1;
            for (int j = 0; j < np1*np1; j++) P[q*np1*np1 + j] = L[qx*np1 + j % np1]*L[qy*np1 + j / np1];
            for (int j = 0; j < np1*n; j++) {
                U[q*np1*n + j] = L[qx*np1 + j % np1]*E[qy*n + j / np1];
                V[q*np1*n + j] = E[qx*n + j % n]*L[qy*np1 + j / n];
            }
            for (int j = 0; j < n*n; j++) W[q*n*n + j] = E[qx*n + j % n]*E[qy*n + j / n];
            Q[q] = quad.w[qx]*quad.w[qy];
        }
        return true;
    }
};

}  // namespace mimsem
