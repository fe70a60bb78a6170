// mimsem_amd/csrc/ctx.hpp -- device context shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "../../include/mimsem_hip.h"
#include "basis_host.hpp"
#include <cstdlib>

// CLOSED EXPERIMENTS: the variants that were built, measured and declined (their records are under profiles/).
//   default build           their switches do not exist (exp_env is a compile-time null) and the kernel families only they launch are not
//                           compiled in (kExperiments: the fused-scatter element kernels, the tile mode of k_apply_wave, the 16-lane order-4
//                           walk of solve_schur_column_3, the residency variants of k_thomas_dpp2, the row-parallel 9 x 9 inverse);
//   -DMIMSEM_WITH_EXPERIMENTS  (scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS" -> build_ab/libmimsem_hip_exp.so, MIMSEM_LIB=...)
//                           everything is compiled in and the switches are read when MIMSEM_EXPERIMENTS=1 is set in the environment -- what
//                           scripts/ab_*.sh and the variants' parity tests run against.
constexpr int MIMSEM_RD_COUNTERS = 8;
#ifdef MIMSEM_WITH_EXPERIMENTS
constexpr bool kExperiments = true;
inline const char* exp_env(const char* name) {
    static const bool on = std::getenv("MIMSEM_EXPERIMENTS") && std::atoi(std::getenv("MIMSEM_EXPERIMENTS")) != 0;
    return on ? std::getenv(name) : nullptr;
}
#else
constexpr bool kExperiments = false;
inline const char* exp_env(const char*) { return nullptr; }
#endif

namespace mimsem {
extern thread_local std::string g_last_hip_error;
int hip_fail(hipError_t e, const char* what);
}

#define MIMSEM_HIP_TRY(call)                                                        \
    do { hipError_t e__ = (call);                                                   \
         if (e__ != hipSuccess) return mimsem::hip_fail(e__, #call); } while (0)

// element sizes for order n (quadrature collocated: m == n)
struct ElemSizes {
    int n, np1, mp1, mp12, n0e, n1e, n2e;
    explicit ElemSizes(int order = 1)
        : n(order), np1(order + 1), mp1(order + 1), mp12((order + 1)*(order + 1)),
          n0e((order + 1)*(order + 1)), n1e((order + 1)*order), n2e(order*order) {}
};

struct mimsem_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t cap_stream = nullptr, cap_saved = nullptr; bool cap_active = false, cap_swapped = false;      // mimsem_graph_begin / _end
    hipStream_t own_stream = nullptr;          // mimsem_ctx_use_own_stream
    void* h_pin = nullptr;                     // 4 KB of pinned host memory: small read-backs (mimsem_memcpy_d2h) skip the pageable-copy path
    ElemSizes es;
    int nEl = 0, nk = 0, n0 = 0, n1 = 0, n2 = 0;
    bool inds2_contig = true;
    mimsem::BasisTables tab;

    // ---- HBM-resident data (layout: DESIGN.md "Data layout in HBM") ----
    double* d_E = nullptr;      // [mp1][n]   edge basis at quad points
    double* d_w = nullptr;      // [mp1]      GLL weights
    double* d_xn = nullptr;     // [np1]      nodal (GLL) points, for Lagrange evaluation at departure points
    double* d_U = nullptr;      // dense tables for the element-matrix kernels
    double* d_V = nullptr;
    double* d_W = nullptr;
    double* d_P = nullptr;
    double* d_J = nullptr;      // [nEl][4][mp12]  component-major per element (coalesced per component)
    double* d_det = nullptr;    // [nEl][mp12]
    double* d_th = nullptr;     // [nk][nEl][mp12] thickness at the element's own quad points
    double* d_tI = nullptr;     // [nk][nEl][mp12] inverse thickness
    bool cheb_pend = false;     // MIMSEM_CHEB_PEND=1 (experiments build): mimsem_block_chebyshev_solve in two launches per step
    bool blocks_mfma = false;   // block pass of the Chebyshev / Richardson sweeps on the matrix cores (default at p = 4, MIMSEM_BLOCKS_MFMA=0|1 at context creation overrides; p <= 3: register-row form)
    double* d_tIp = nullptr;    // [2][nk/2 + 1][nEl][mp12][2]: the same in level PAIRS {L, L+1}, first index = parity of L (k_apply_wave: one 16-byte load per two levels)
    double* d_tIn = nullptr;    // [2][nk/2 + 1][n0][2]: the same per NODE (MIMSEM_WAVE_TNODE=1; null unless every element holds the same value at a shared node)
    bool have_levels = false;
    int* d_i0 = nullptr;        // [nEl][n0e]
    int* d_i1x = nullptr;       // [nEl][n1e]
    int* d_i1y = nullptr;       // [nEl][n1e]
    int* d_i2 = nullptr;        // [nEl][n2e] or null (contiguous)
    int* d_iq = nullptr;        // [nEl][mp12] quad-grid slots (projection operators), may be null
    int nq = 0;
    // deterministic scatter-add plans: vector slot -> up to K element-local result slots (-1 = none)
    int* d_g1 = nullptr;        // [n1][2]   into ye1[e*2*n1e + j]  (j<n1e: x edge, else y edge)
    int* d_bplan = nullptr;     // [nEl][2 n1e][4] = {slot, d_g1[slot][0], d_g1[slot][1], 0}: the block passes' view of d_g1, one 16-byte load per block row
    int* d_g0 = nullptr;        // [n0][G0]  into ye0[e*n0e + j]
    int G0 = 4;
    // fused scatter-add of 1-form results: element groups = workgroups, group-local slot ids
    bool fused1 = false;
    int f_ngroups = 0, f_lmax = 0, f_nps = 0, f_npart = 0;
    int* d_fperm = nullptr;             // [ngroups][EPB] element of each lane group (-1 = padding)
    unsigned short* d_flid = nullptr;   // [ngroups][lmax][2] positions (el*2*n1e + dof) of the 1-2 contributions of each local slot
    int* d_fslot = nullptr;             // [ngroups][lmax] vector slot (>=0, complete in group) or -(partial index+1)
    int* d_fcnt = nullptr;              // [ngroups] local ids in use
    int* d_pslot = nullptr;             // [nps] perimeter slots
    int* d_ppart = nullptr;             // [nps][2] their partial-sum indices (-1 = none)
    // wave-level fused scatter-add (k_apply_wave, elem_wave.inc): wave-groups of 64/LPE neighbouring elements; tables: build_wave_plan
    bool wave1 = false;
    int wave_order = 3;                 // bit 0: XCD-contiguous block order, bit 1: group-major items (MIMSEM_WAVE_ORDER)
    int wave_lch = 0;                   // MIMSEM_WAVE_LCH override of the levels per chunk
    int wave2_mode = 1;                 // MIMSEM_WAVE2: 2-form-valued operators on k_apply_wave2 (p = 3): 0 none, 1 Whmat / WtQUmat / WtQdUdz, 2 also Wmat
    int wave_cpp = 0;                   // MIMSEM_WAVE_CPP override of the chunks per work item (0: heuristic)
    int w_ngroups = 0, w_nsing = 0, w_nps = 0, w_npart = 0, w_ndirect = 0;
    int w_npwritten = 0;                // partial sums a level really gets (w_npart is the padded row: w_nsides x 2 x 16)
    // in-kernel completion of the perimeter (round 3, elem_wave.inc "finishing phase"): the partial sums are laid out per SIDE (the
    // slots two wave-groups share), and the group that arrives second at a side's counter finishes its slots -- no second launch
    bool w_fin = false;                 // the plan supports it (MIMSEM_WAVE_FIN=0 keeps the perimeter pass)
    int w_nsides = 0; int w_partmem = 0;      // w_partmem: 0 uncached, 1 fine-grained, 2 plain (MIMSEM_WPART_MEM, experiments)
    int4* d_wfin = nullptr;             // [w_ngroups][8] {side, arrivals that complete it (2; 1: a side of the group's own), entries, 0} or {-1, 0, 0, 0}
    int* d_wsslot = nullptr;            // [w_nsides][16] y slot of entry j of the side (-1: none)
    int* d_wcnt = nullptr;              // [w_nsides][nk] arrival counters, zero between launches (the finishing wave resets its own)
    double* d_wpart = nullptr;          // [nlev][w_npart + 128] partial sums of the finishing phase: UNCACHED device memory (plain stores go
    long long wpart_doubles = 0;        //   through to the memory side: visible to a finishing wave on another XCD once acknowledged)
    int ensure_wpart(long long doubles);
    // a split apply (mimsem_op_apply_part): its partial sums live in a buffer no other entry point uses, and the pending BOUNDARY part
    // is remembered so that only the matching INTERIOR part can consume it
    double* d_wsplit = nullptr; long long wsplit_doubles = 0;
    int ensure_wsplit(long long doubles);
    struct { bool pending = false; int op = 0, lev0 = 0, nlev = 0; unsigned flags = 0; const double* y = nullptr; long long ys = 0; } split;
    int w_nbgroups = 0, w_nbrec = 0; bool w_split = false;     // interior / boundary split (mimsem_ctx_set_halo_slots): boundary prefix sizes
    std::vector<int> h_i1x, h_i1y, h_i0; std::vector<double> h_J, h_det;      // host copies of the mesh for re-deriving the plan
    std::vector<int> h_e0;              // element -> 0-form slot lists (node multiplicities, likewise)
    std::vector<int> h_e1x, h_e1y;      // element -> 1-form slot lists, kept on EVERY context (edge multiplicities of the PCBJACOBI builders, ksp.hip)
    int4* d_wlane = nullptr;            // [w_ngroups][64] {element of the lane, load pair: even slot b, staging positions of x[b] and of x[b+1]
                                        //   (two 16-bit positions each; the dump position where nobody wants the value)}
    int4* d_wtfin = nullptr; int w_ntiles = 0, w_ninner = 0;      // tile mode (round 5): [w_ntiles][64] {slot, LDS position of part A, of part B, 0}
    int4* d_wplan = nullptr;            // [w_ngroups][64] store pair {dst, result positions of its first and second slot (2 x 16 bit, the
                                        //   strip's zero for a missing contributor), 0}: dst >= 0: y[dst], y[dst+1]; dst <= -2: partial sums
                                        //   -(dst+2), +1 of the workspace row (unused lanes: its dump tail)
    int2* d_wsing = nullptr;            // [w_ngroups][64] optional 8-byte store round {slot or -1, positions} (MIMSEM_WAVE_SINGLES=1)
    int* d_wnode = nullptr;             // [w_ngroups][64] node slot of the lane's quadrature point (RotMat's vorticity)
    double* d_wG = nullptr;             // [w_ngroups][64][4] {gaa, gab, gbb, 1/det} of the lane's point: Q/det J^T J, in WAVE-GROUP order
    double* d_wR = nullptr;             // [w_ngroups][64]    (-J00 J11 + J01 J10) Q/det: RotMat
    int4* d_wprec = nullptr;            // [w_nps] {slot, partial 0, partial 1 (-1: none), 0}: slots finished by k_wave_perim (and slots no element
                                        //   touches: no partial at all, written as 0)
    // workspace
    double* d_ye = nullptr;     // [nk_ws][nEl][max(2*n1e, n0e)] element-local results
    long long ye_doubles = 0;
    double* d_cheb = nullptr; long long cheb_doubles = 0;      // mimsem_block_chebyshev_solve: the second iterate and two direction vectors, [3][nlev][n1]
    int ensure_cheb(long long doubles);
    int *d_d0 = nullptr, *d_d1x = nullptr, *d_d1y = nullptr;   // direct-write slots (single-contributor DoFs), see ElemArgs
    int *d_sh0 = nullptr, *d_sh1 = nullptr; int nsh0 = 0, nsh1 = 0;   // slots with >= 2 contributors: the only ones pass 2 visits
    bool direct = false;
    bool colstat_valid = false;
    double* d_colratio = nullptr;   // [nEl] |last correction| / |solution| of that solve
    int* d_colstat = nullptr;   // [1 + nEl] status of the last block-tridiagonal column solve (mimsem_column_solve_status)
    double* d_col = nullptr;    // column-solver workspace
    unsigned* d_rdcnt = nullptr;   // [MIMSEM_RD_COUNTERS] arrival counters of the one-launch rowdot (krylov_kernels.hip: k_rowdot_fused, calls of up to 8 rows), zero between calls
    std::vector<void*> graphs;  // the recordings (mimsem_graph*) made on this context and still alive: orphaned by mimsem_ctx_destroy
    bool rd_two = false;           // MIMSEM_ROWDOT_TWO (experiments build): the two-launch rowdot, for the A/B of round 6
    bool memset_node = false; int blu_stop = 0;      // MIMSEM_MEMSET_NODE / MIMSEM_BLU_STOP, read once at creation
    std::vector<char> h_halo1;  // [n1] 1 = the 1-form slot takes part in a halo exchange (mimsem_ctx_set_halo_slots): its second element lives on another rank
    int* d_forceflag = nullptr; int n_forceflag = 0;      // mimsem_column_flag_for_test: columns the next solve treats as flagged (one-shot)
    int pivot_fallback = 1;     // mimsem_column_set_pivot_fallback: flagged columns are re-solved by a band LU with partial pivoting (column_pivot.inc)
    double* d_lu = nullptr;     // its workspace: per wavefront of the fallback's grid, the rows of U and the multipliers of one column
    long long lu_doubles = 0;
    long long col_doubles = 0;
    double col_param = 0.0;             // scalar argument of the *_ex column operators (dt_fric / dt)
    const double* col_uh = nullptr;     // horizontal velocity [nk][n1] of the *_up column operators
    long long col_uhs = 0;
    long long bytes = 0;
    int swz = 1;                   // MIMSEM_NOSWZ=1 disables the XCD-aware block order (tuning)
    int lch_override = 0;          // MIMSEM_LCH environment override (tuning)
    // measurement hook: event triples (start, mid, end) around pass 1 / pass 2 of mimsem_op_apply
    bool profiling = false;
    int prof_every = 1; long long prof_count = 0;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    hipEvent_t next_event();
    hipEvent_t ev_k1[2] = {nullptr, nullptr};   // start/stop events of the next element kernel (null = not profiling)
    hipEvent_t ev_k2[2] = {nullptr, nullptr};   // ... of the next gather-sum kernel
    std::vector<char> ev_has2;                  // per profiled apply: a second kernel recorded its pair (pooled events keep OLD stamps otherwise)
    void mark_k2() { if (ev_k2[0] && !ev_has2.empty()) ev_has2.back() = 1; }

    std::vector<void*> retired;          // outgrown workspaces (still referenced by captured graphs), freed with the context
    bool is_capturing() const;           // the context's stream is inside a hipGraph capture (workspaces must not grow there)
    int ensure_ye(long long doubles);
    int ensure_col(long long doubles);
    int gs_fused = -1;                  // second Gram-Schmidt pass + normalisation in two launches (mimsem_krylov_gs_control; -1: from the environment)
    int* gs_flag = nullptr;             // caller's word (device or pinned host), set to 1 when that form's Pythagorean norm would cancel
    double* d_kry = nullptr;            // partial sums of the Krylov multi-dot
    long long kry_doubles = 0;
    int ensure_kry(long long doubles);
};

// tile mode of the wave-level kernel (round 5): finishing entries per tile, doubles of a tile's LDS row, levels the LDS rows hold
constexpr int MIMSEM_WTF = 64, MIMSEM_WTP = 64, MIMSEM_WTLEV = 32;

// kernels (elem_kernels.hip / column_kernels.hip) ---------------------------------------------------
struct ElemArgs {
    int nEl, nlev, lev0, total;
    int swz;                       // XCD-aware work-item order on/off
    int lch;                       // levels handled by one work item (level-invariant data stays in registers)
    unsigned flags;
    double scale, alpha;
    const double *J, *det, *tI, *th, *E, *w;
    const double* tIp; int tnp;          // level-pair copy of tI, pairs per parity
    long long tps; int tnode;            // doubles between two pair rows of tIp; tnode: tIp is the NODAL table (entry = node slot of the point, wnode) instead of [element][point]
    const int *i0, *i1x, *i1y, *i2, *iq;
    const double* f; long long fs;
    const double* f2; long long f2s;   // second coefficient field (velocity of the upwinded operators)
    double param;                       // tau of the upwinded operators
    const double* xn;                   // nodal points
    const double* x; long long xs;
    double* out; long long os;     // element-local results (or the 2-form output vector itself); fused: partial sums
    // fused 1-form scatter-add
    const int* fperm; const unsigned short* flid; const int* fslot; const int* fcnt; int ngroups, lmax;
    double* y; long long ys; int accum;
    // wave-level fused scatter-add (k_apply_wave)
    const int4* wlane; const int4* wplan; const int2* wsing; const int* wnode; const double* wG; const double* wR; int wgroups; int wg0; int wdump; int wcpp;
    const int4* wfin; const int* wsslot; int* wcnt;      // finishing phase (null: the perimeter pass follows)
    const int4* wtfin; int wtile;                        // tile mode: finishing entries per tile (stride wtile), or null
    int wfence;                      // finishing phase, experiment: partial sums in PLAIN memory, one agent-scope release fence per wavefront before its arrival
    double Etab[20];                 // edge-basis table E[mp1][n] by value (orders <= 4): SGPRs, no load in the kernel
    double Wq[5];                    // GLL weights by value (orders <= 4)
    long long* wstamps;              // diagnostic build (MIMSEM_STAMPS): 16 s_memtime stamps per work item, else null
    // direct path: DoFs touched by exactly ONE element are written straight into y (no ye round trip, no pass 2 for them)
    const int *d0, *d1x, *d1y;       // [nEl][n0e|n1e]: the slot when the element is its only contributor, else -1 (null = off)
};

// The vector update a Chebyshev step still owes (mimsem_block_chebyshev_solve: the gather epilogue of step k folded into the element pass of
// step k + 1).  For the slots of its element a lane forms  z = sum of the element-local P r through the plan (the epilogue's order),
// p = z + beta p_in,  x = x_in + alpha p  and applies the operator to THAT x; the slot's first contributor stores x and p into the OTHER
// pair of buffers (the other elements of the slot read the old values in this same launch).  first: x_in = 0 and p_in are not read.
struct ElemPending {
    const int* plan; const double* ze; long long zes;
    double alpha, beta; int first;
    const double* p_in; double* p_out; long long ps;                     // [nlev][n1] rows (x_in is ElemArgs::x, stride ElemArgs::xs)
    double* x_out; long long xos;
    double* upd; long long us;                                           // receives z if given
};
// pass 2 with an epilogue (the Richardson sweeps): what happens to the gathered sum `acc` of a slot
struct GatherEpilogue {
    int mode;                        // 1, 5: d = dinv*(b - acc) ; 2, 3: d = acc.   1, 2: x += d.   3, 5: p = d + beta p ; x += alpha p.   upd = d if given
    const double* b; long long bs;
    const double* dinv; long long ds;
    double* upd; long long us;
    double alpha = 1.0, beta = 0.0;  // mode 3 (Chebyshev semi-iteration)
    double* p = nullptr; long long ps = 0;
    const double* escale = nullptr; long long ess = 0;      // block pass: per (level, element) factor of the element blocks
    // mode 4 (round 5): the vector algebra of a Chebyshev step on B = P A with acc = (B d)[s]:  x += d;  r -= acc;  d = alpha d + beta r.
    // d lives in p (in/out), r in cr
    double* cr = nullptr; long long crs = 0;
    int zero = 0;                    // modes 3, 5, first step of a solve from x = 0: p = d, x = alpha d -- neither is read (mimsem_block_chebyshev_solve, mimsem_sw_dual_chebyshev)
    int noacc = 0;                   // ... and the operator result is zero: acc = 0, ye is not read (mode 5 without a block pass in between)
};
// two independent Chebyshev sweeps in the same launches (elem_kernels.hip: k_sw_pair): the block pass and the gather epilogue of one level
struct PairBlocks { int nEl, lch; const int *i1x, *i1y, *plan; const double *B, *ye; long long yes; const double* b; double* ze; long long zes; const int4* bplan; };
struct PairGather { const double* ye; long long yes; const int* plan; int nslots; GatherEpilogue g; double* x; };
// phase PA of the 1-form mass sweep (0 element pass, 1 block pass, 2 gather epilogue) and phase PB of the upwinded 0-form sweep (0 element
// pass, 1 gather epilogue) in ONE launch; orders 2..4
int launch_sw_pair(mimsem_ctx* c, int PA, int PB, const ElemArgs& ea, const PairBlocks& ba, const PairGather& ga, const ElemArgs& eq, const PairGather& gq);
int launch_gather_epilogue(mimsem_ctx* c, int form, int nlev, const double* ye, long long ye_stride, const GatherEpilogue& g,
                           double* x, long long xs);
int launch_blocks_residual(mimsem_ctx* c, int nlev, const double* B, const double* ye, long long yes,
                           const double* b, long long bs, double* ze, long long zes,
                           const double* escale = nullptr, long long ess = 0);
int launch_elem_apply(mimsem_ctx* c, int op, const ElemArgs& a);
int launch_elem_apply_pending(mimsem_ctx* c, const ElemArgs& a, const ElemPending& pd);      // Umat with the owed Chebyshev update (orders <= 5)
int launch_gather_perim(mimsem_ctx* c, int nlev, const double* yp, long long yps, int accum, double* y, long long ys,
                        const int* pslot = nullptr, const int* ppart = nullptr, int nps = -1);
int launch_apply_wave(mimsem_ctx* c, int op, const ElemArgs& a);
int launch_apply_wave2(mimsem_ctx* c, int op, const ElemArgs& a);
int launch_wave_perim(mimsem_ctx* c, int nlev, const double* yp, long long yps, int accum, double* y, long long ys, int r0 = 0, int r1 = -1);
int launch_gather_sum(mimsem_ctx* c, int form, int nlev, const double* ye, long long ye_stride, int accum,
                      double* y, long long ys, bool shared_only = false);
int launch_blocks_apply(mimsem_ctx* c, int form, int nlev, int transposed, const double* B, long long bstride_lev,
                        const double* x, long long xs, double* y, long long ys, double alpha, int accum,
                        const double* escale = nullptr, long long escale_stride = 0);
int launch_halo_segments(mimsem_ctx* c, const int* idx, int nseg, const int* seg_off, int s_begin, int s_end, int nlev, int mode,
                         double* buf, double* v, long long vs);
int launch_elmats(mimsem_ctx* c, int op, int lev, double scale, unsigned flags, const double* f, double* out,
                  const double* f2 = nullptr, double param = 0.0);
int launch_incidence(mimsem_ctx* c, int which, int nlev, const double* x, long long xs, double* y, long long ys);
int launch_interp_quad(mimsem_ctx* c, int form, int global, int nlev, const double* x, long long xs, double* out, long long os);
int launch_sw_operator(mimsem_ctx* c, int nlev, double a, double grav, double H, const double* f0, long long f0s,
                       const double* x, long long xs, double* y, long long ys);
int launch_sw_blocks_apply(mimsem_ctx* c, int nlev, const double* B, const double* x, long long xs, double* y, long long ys);
int launch_sw_operator_precond_chebyshev(mimsem_ctx* c, int nlev, double a, double grav, double H, const double* f0, long long f0s,
                                         const double* B, double ca, double cb, double* x, long long xs, double* r, long long rs, double* d, long long ds);
int launch_sw_chebyshev_step2(mimsem_ctx* c, int nlev, double a, double grav, double H, const double* f0, long long f0s, const double* B,
                              int pending, double pca, double pcb, double ca, double cb, double* x, long long xs,
                              const double* r_in, const double* d_in, double* r_out, double* d_out, double* rh, double* dh, long long vs);
int launch_sw_chebyshev_flush(mimsem_ctx* c, int nlev, double ca, double cb, double* x, long long xs, double* r, double* d, long long vs);
int launch_sw_operator_precond(mimsem_ctx* c, int nlev, double a, double grav, double H, const double* f0, long long f0s,
                               const double* B, const double* x, long long xs, double* z, long long zs,
                               const double** unassembled = nullptr /* nlev == 1: skip the 1-form gather of z, return the element-local results */);
int launch_halo_pack(mimsem_ctx* c, const int* idx, int count, int nlev, const double* v, long long vs, double* buf);
int launch_halo_unpack(mimsem_ctx* c, const int* idx, int count, int nlev, int mode, const double* buf, double* v, long long vs);

// column_kernels.hip internals used by api.hip
int mimsem_block_inverse_inplace(mimsem_ctx* c, long long nblocks, int n, double* blocks, int* err = nullptr);
int mimsem_colop_block_inverse_apply(mimsem_ctx* c, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                                     const double* f, long long fs, const double* x, long long xs,
                                     double* y, long long ys, double alpha);
