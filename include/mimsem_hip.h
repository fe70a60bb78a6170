/*
 * include/mimsem_hip.h -- C ABI of libmimsem_hip.so: the MI355X (gfx950) element-operator engine
 * for the MiMSEM per-element assemble/apply layer and the per-column (HEVI) vertical operators.
 *
 * Drop-in boundary.  The reference has no FFI; its boundary is the C++ class surface
 * of eul/Assembly.h, eul/VertOps.h, eul/L2Vecs.h over PETSc Vec/Mat.  Each entry point below names
 * the reference interface it replaces; INTEGRATION.md shows the reference-side binding.  The
 * reference idiom
 *       X->assemble(<fields>, lev, scale, ...);  MatMult(X->M, x, y);        (e.g. eul/HorizSolve.cpp:216-221)
 * becomes ONE call  mimsem_op_apply(ctx, MIMSEM_OP_X, ...)  -- matrix-free, batched over levels --
 * and callers that still need the assembled PETSc Mat get the dense element blocks that the
 * reference hands to MatSetValues from  mimsem_op_element_matrices().
 *
 * Conventions: plain pointers and sizes only.  All arithmetic IEEE FP64, all indices int32.
 * "dev" pointers are device (HBM) addresses (hipMalloc / torch tensor data_ptr / mimsem_malloc);
 * "host" pointers are ordinary memory.  Vectors follow the PETSc layout the reference reads through
 * VecGetArray: contiguous doubles; a multi-level field is addressed as base + k*lev_stride.
 * Every function returns 0 on success or a negative MIMSEM_ERR_* code (the reference itself has no
 * error convention: PETSc codes are ignored and Inv's error int is dropped, eul/VertOps.cpp:434).
 * Launches are asynchronous on the context's stream; nothing here synchronises unless it says so.
 */
#ifndef MIMSEM_HIP_H
#define MIMSEM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MIMSEM_ABI_VERSION 1

enum {
    MIMSEM_OK = 0,
    MIMSEM_ERR_ARG = -1,          /* null / out-of-range argument                              */
    MIMSEM_ERR_UNSUPPORTED = -2,  /* order outside 1..7 or quadrature order != element order   */
    MIMSEM_ERR_HIP = -3,          /* a HIP runtime call failed (see mimsem_last_hip_error)     */
    MIMSEM_ERR_STATE = -4,        /* call needs data not yet given (e.g. levels not set)       */
    MIMSEM_ERR_SINGULAR = -5      /* a block inverse hit |pivot| < 1e-12 (LinAlg.cpp:243)      */
};

typedef struct mimsem_ctx mimsem_ctx;

/* What one device holds: any set of cubed-sphere patches flattened to element -> vector-slot tables.
 * Host pointers, copied at creation.  Replaces the per-rank Topo/Geom members the operator classes
 * read (eul/Topo.h:5-51, eul/Geom.h:8-36): pass Topo::elInds*_l for a single reference rank, or
 * device-global slots (loc* composed with a compaction) when several patches share one GPU. */
typedef struct mimsem_mesh_desc {
    int elOrd;            /* Topo::elOrd                                                        */
    int quadOrd;          /* Geom::quad->n  (must equal elOrd)                       */
    int nEl;              /* elements on this device                                            */
    int nk;               /* Geom::nk vertical levels (>= 1; 1 for the shallow-water stack)     */
    int n0, n1, n2;       /* lengths of 0/1/2-form vectors per level (Topo::n0,n1,n2)           */
    const int* inds0;     /* [nEl][(n+1)^2]  Topo::elInds0_l   eul/Topo.cpp:200-212             */
    const int* inds1x;    /* [nEl][(n+1)n]   Topo::elInds1x_l  :214-226                         */
    const int* inds1y;    /* [nEl][n(n+1)]   Topo::elInds1y_l  :228-240                         */
    const int* inds2;     /* [nEl][n^2] or NULL = element-contiguous e*n^2+i (Topo::elInds2_l :242-251) */
    const double* det;    /* [nEl][(m+1)^2]      Geom::det   eul/Geom.cpp:726-741               */
    const double* J;      /* [nEl][(m+1)^2][4]   Geom::J  as J00 J01 J10 J11                    */
    const double* thick;    /* [nk][nEl][(m+1)^2] layer thickness AT each element's quad points
                               (Geom::thick[k][Geom::elInds0_l(e)[q]]), or NULL -> 1            */
    const double* thickInv; /* same shape, Geom::thickInv; NULL -> 1/thick (or 1)               */
    const int* indsq;     /* [nEl][(m+1)^2] Geom::elInds0_l (eul/Geom.cpp:799-811): slots of the quad-point
                             grid vectors the projection operators read; NULL if they are not used */
    int nq;               /* length of a quad-point grid vector (Geom::n0)                        */
} mimsem_mesh_desc;

/* horizontal operator classes, eul/Assembly.h (src/Assembly.h twins: scale=1, no thickness) */
enum mimsem_op {
    MIMSEM_OP_UMAT = 0,     /* Umat::assemble(lev,scale,vert_scale)         Assembly.cpp:66-153    1-form -> 1-form */
    MIMSEM_OP_WMAT = 1,     /* Wmat::assemble(lev,scale,vert_scale)         :324-373               2 -> 2           */
    MIMSEM_OP_UHMAT = 2,    /* Uhmat::assemble(h2,lev,const_vert,scale)     :416-474   f=h2 (2-form)  1 -> 1        */
    MIMSEM_OP_PMAT = 3,     /* Pmat::assemble(lev,scale)                    :2004-2046             0 -> 0           */
    MIMSEM_OP_PHMAT = 4,    /* Pmat::assemble_h(lev,scale,h2)               :2048-2098 f=h2           0 -> 0        */
    MIMSEM_OP_WTQUMAT = 5,  /* WtQUmat::assemble(u1,lev,scale)              :933-986   f=u1 (1-form)  1 -> 2        */
    MIMSEM_OP_ROTMAT = 6,   /* RotMat::assemble(q0,lev,scale)               :1030-1083 f=q0 (0-form)  1 -> 1        */
    MIMSEM_OP_WHMAT = 7,    /* Whmat::assemble(rho,lev,scale,vert_scale_rho):1243-1299 f=rho (2-form) 2 -> 2        */
    MIMSEM_OP_UTMAT = 8,    /* Ut_mat::assemble(lev,scale)                  :1338-1386 (needs lev+1 < nk) 1 -> 1    */
    MIMSEM_OP_UTMAT_H = 9,  /* Ut_mat::assemble_h(lev,scale,rho)            :1388-1438 f=rho          1 -> 1        */
    MIMSEM_OP_UTQWMAT = 10, /* UtQWmat::assemble(u1,scale)                  :1490-1538 f=u1           2 -> 1        */
    MIMSEM_OP_WTQDUDZ = 11, /* WtQdUdz_mat::assemble(u1,scale)              :1581-1640 f=u1           1 -> 2        */
    MIMSEM_OP_WMATINV = 12, /* WmatInv::assemble(lev,scale)                 :1673-1722 element-wise inverse 2 -> 2  */
    MIMSEM_OP_WHMATINV = 13,/* WhmatInv::assemble(rho,lev,scale)            :1744-1802 f=rho          2 -> 2        */
    /* upwinded operators of the shallow-water stack (src/ flavour: scale = 1, no thickness), applied through
     * mimsem_op_apply_up: f = the op's field, u = local 1-form velocity that defines the departure points */
    MIMSEM_OP_PHMAT_UP = 14, /* Phmat::assemble_up(ul,hl,fac,dt)  src/Assembly.cpp:499-567   f=hl (2-form)  0 -> 0 */
    MIMSEM_OP_ROTMAT_UP = 15,/* RotMat_up::assemble(q0,ul,fac,dt) src/Assembly.cpp:1784-1853 f=q0 (0-form)  1 -> 1 */
    /* eul-flavour operators whose TEST functions are evaluated at departure points (mimsem_op_apply_up) */
    MIMSEM_OP_UMAT_UP = 19,   /* Umat::assemble_up(lev,scale,tau,ui,uj)      Assembly.cpp:156-279  f=ui, u=uj (local 1-forms) 1 -> 1 */
    MIMSEM_OP_UHMAT_UP = 20,  /* Uhmat::assemble_up(h2,lev,scale,dt,u1)      :477-560              f=h2, u=u1, tau=dt        1 -> 1 */
    MIMSEM_OP_UVEC_HU_UP = 21,/* Uvec::assemble_hu_up(lev,scale,vel,rho,fac,tau,vel2) :2281-2373   x=vel, f=rho, u=vel2, alpha=fac */
    /* Held-Suarez boundary-layer friction (mimsem_op_apply_up: f = exner at the level, u = exner at level 0 with
     * u_stride = 0, tau = dt; both 2-forms in the horizontal layout) */
    MIMSEM_OP_UMAT_RAY = 22,  /* Umat_ray::assemble(lev,scale,dt,exner_k,exner_s) Assembly.cpp:1876-1979    1 -> 1 */
    /* projections from the quadrature-point grid (initial conditions, Coriolis): x lives on the quad grid */
    MIMSEM_OP_WTQ = 16,      /* WtQmat::assemble  Assembly.cpp:707-751   quad scalar -> 2-form                        */
    MIMSEM_OP_PTQ = 17,      /* PtQmat::assemble  :766-808               quad scalar -> 0-form                        */
    MIMSEM_OP_UTQ = 18,      /* UtQmat::assemble  :824-902               quad vector [nq][2] -> 1-form                */
    MIMSEM_OP_COUNT
};
/* op flag: the boolean the reference method takes (vert_scale / const_vert / vert_scale_rho) */
#define MIMSEM_FLAG_VERT   1u
/* y += result instead of y = result (Uvec::assemble_hu(..., zero_and_scatter=false, ...), :2198-2279) */
#define MIMSEM_FLAG_ACCUM  2u
/* apply the transpose (mimsem_elem_blocks_apply: blocks are read column-major, the coalesced direction) */
#define MIMSEM_FLAG_TRANSPOSE 4u

/* ---- context ------------------------------------------------------------------------------- */
int  mimsem_abi_version(void);
/* 1 when the library was built with -DMIMSEM_WITH_EXPERIMENTS (the closed experiments' kernel variants and their MIMSEM_* switches are compiled
 * in: scripts/build_variant.sh exp "-DMIMSEM_WITH_EXPERIMENTS"), 0 for the default build.  Test / A-B infrastructure; replaces nothing. */
int  mimsem_build_has_experiments(void);
const char* mimsem_strerror(int code);
const char* mimsem_last_hip_error(void);
int  mimsem_device_count(void);

/* device = HIP ordinal.  Builds GLL/edge tables (eul/Basis.cpp), copies the mesh, builds the
 * deterministic scatter-add plans.  Replaces: Topo+Geom+LagrangeNode/Edge ctor arguments of every
 * Assembly.h class (eul/Assembly.cpp:26-30). */
int  mimsem_ctx_create(const mimsem_mesh_desc* desc, int device, mimsem_ctx** out);
void mimsem_ctx_destroy(mimsem_ctx* ctx);
/* run all subsequent launches on this hipStream_t (NULL = default stream) */
int  mimsem_ctx_set_stream(mimsem_ctx* ctx, void* hip_stream);
int  mimsem_ctx_sync(mimsem_ctx* ctx);
/* refresh thickness after Geom::initTopog (eul/Geom.cpp:743-764); host arrays [nk][nEl][(m+1)^2] */
int  mimsem_ctx_set_levels(mimsem_ctx* ctx, const double* thick, const double* thickInv);
/* bytes of device workspace the context currently holds */
long long mimsem_ctx_workspace_bytes(const mimsem_ctx* ctx);
/* Levels one work item of the element kernel (pass 1 of mimsem_op_apply) keeps its level-invariant data (metric, determinant,
 * gather slots) in registers for, at a call over nlev levels: the byte model of bench.py's roofline needs it (the metric of an
 * element is re-read once per chunk, not once per level). */
int  mimsem_op_level_chunk(const mimsem_ctx* ctx, int nlev);
/* The wave-level fused form of the 1-form -> 1-form operators (Umat, Uhmat, RotMat, Ut_mat; the default for orders <= 4 unless
 * MIMSEM_WAVE=0): out[0] = wave-groups (wavefronts per level chunk), out[1] = vector slots the element kernel writes straight
 * into y, out[2] = partial sums per level it leaves in the workspace, out[3] = slots the perimeter pass finishes,
 * out[4] = levels one wavefront works through at a call over nlev levels (chunks of 8, several per wavefront when the launch has wavefronts to spare).  Returns 0 when the form is off (all five are then 0). */
int  mimsem_op_wave_stats(const mimsem_ctx* ctx, int nlev, int out[5]);
/* Interior / boundary split of a 1-form operator apply, so that a halo exchange overlaps the interior work (the
 * reference's MatMult + VecScatterBegin/End, eul/Assembly.cpp:2194-2195): tell the context once which 1-form slots take part in an
 * exchange (ghosts and mirrors of mimsem_halo_create's lists) -- the element groups touching them are moved to the front of the
 * plan -- then run   mimsem_op_apply_part(..., MIMSEM_PART_BOUNDARY);  mimsem_halo_begin(...);
 *                    mimsem_op_apply_part(..., MIMSEM_PART_INTERIOR);  mimsem_halo_end(...).
 * After the BOUNDARY part every marked slot of y holds its complete local sum; the INTERIOR part writes the remaining slots.
 * Operators / orders without the wave-level form (and contexts without marked slots) run whole in the BOUNDARY part and do nothing
 * in the INTERIOR part, so the sequence above is always valid.  MIMSEM_PART_ALL = mimsem_op_apply.  Set-up call: not inside a
 * stream capture (MIMSEM_ERR_STATE), invalidates nothing a captured graph holds (old tables are retired, not freed).
 * CONTRACT of a split apply: the BOUNDARY part leaves partial sums for the INTERIOR part in a buffer of the context that only split
 * applies use (no other entry point touches it), and the context remembers the pending part {op, levels, y, stride}.  Until the
 * matching INTERIOR part has run, a second BOUNDARY part, or an INTERIOR part with another op / level range / y, returns
 * MIMSEM_ERR_STATE and changes nothing; any other call (whole applies, block / Krylov / column calls) may run in between.
 * mimsem_op_apply_part_reset forgets a pending part (error recovery of the host).                                              */
#define MIMSEM_PART_ALL      0
#define MIMSEM_PART_BOUNDARY 1
#define MIMSEM_PART_INTERIOR 2
int  mimsem_ctx_set_halo_slots(mimsem_ctx* ctx, int form, const int* slots, int n);
int  mimsem_op_apply_part(mimsem_ctx* ctx, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                          const double* f, long long f_stride, const double* x, long long x_stride,
                          double* y, long long y_stride, double alpha, int part);
int  mimsem_op_apply_part_reset(mimsem_ctx* ctx);
/* Measurement hook (bench.py): when on = n > 0, every n-th mimsem_op_apply brackets its element kernel (pass 1) and its
 * gather-sum kernel (pass 2) with hipEvents on the context's stream.  mimsem_ctx_profile_read waits for
 * the stream, returns the accumulated kernel milliseconds and launch count since the last read, resets. */
int  mimsem_ctx_set_profiling(mimsem_ctx* ctx, int on);
int  mimsem_ctx_profile_read(mimsem_ctx* ctx, double* ms_pass1, double* ms_pass2, long long* launches);

/* plain device memory helpers so a C/C++ host (PETSc VecGetArray side) needs no HIP headers */
/* A stream of the context's own (non-blocking), for hosts without HIP headers that want their launches out of the legacy default stream;
 * mimsem_ctx_set_stream still overrides it.  MIMSEM_ERR_STATE while a graph is being recorded. */
int  mimsem_ctx_use_own_stream(mimsem_ctx* ctx);
int  mimsem_malloc(void** dev, long long bytes);
int  mimsem_free(void* dev);
int  mimsem_memcpy_h2d(mimsem_ctx* ctx, void* dev, const void* host, long long bytes);
int  mimsem_memcpy_d2h(mimsem_ctx* ctx, void* host, const void* dev, long long bytes);
int  mimsem_memset(mimsem_ctx* ctx, void* dev, int byte, long long bytes);

/* ---- horizontal operators (rows B1..B17) ---------------------------------------------------- */
/* y_k = A_op(geometry, level k, f_k) x_k   for k = lev0 .. lev0+nlev-1, all on device vectors with
 * the "local" (ghost-in-place) layout.  Replaces X->assemble(...) + MatMult(X->M, x, y) for the
 * enum's class, and Uvec::assemble/assemble_hu/assemble_wxu (Assembly.cpp:2124-2430), which are
 * UMAT/UHMAT/ROTMAT applied to the velocity (fac -> `alpha`).
 *   f, f_stride : coefficient field of the op (NULL when it takes none), level stride in doubles
 *   x, y        : input / output device vectors, sizes by the op's spaces, level strides in doubles
 *   alpha       : result is scaled by alpha (1.0 for MatMult semantics; Uvec's `fac`)
 *   geom_lev0   : geometry level of the first vector level (normally == lev0; the shallow-water
 *                 stack passes 0 with nk = 1)                                                   */
int mimsem_op_apply(mimsem_ctx* ctx, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                    const double* f, long long f_stride,
                    const double* x, long long x_stride,
                    double* y, long long y_stride, double alpha);

/* Upwinded variants.  PHMAT_UP / ROTMAT_UP: the 0-form (trial function resp. vorticity) is evaluated at the departure
 * points x_q - tau*u_local(x_q) (the reference's tau = 1/(1/(fac*dt)) is formed by the caller).  UMAT_UP / UHMAT_UP /
 * UVEC_HU_UP: the test functions are evaluated at x_q + (op-specific shift).  u: second field (velocity) per level. */
int mimsem_op_apply_up(mimsem_ctx* ctx, int op, int geom_lev0, int nlev, double scale, double tau, unsigned flags,
                       const double* f, long long f_stride, const double* u, long long u_stride,
                       const double* x, long long x_stride, double* y, long long y_stride, double alpha);

/* The dense element blocks the reference passes to MatSetValues (row-major, block order as in the
 * reference: UMAT-like [4][n1e][n1e] = UtQU UtQV VtQU VtQV; ROTMAT [2] = UtQV VtQU; WTQU-like [2];
 * UTQW [2]; 2-form [n2e][n2e]; 0-form [n0e][n0e]).  out: device, [nEl][esz].                     */
int mimsem_op_elmat_size(const mimsem_ctx* ctx, int op);
int mimsem_op_element_matrices(mimsem_ctx* ctx, int op, int geom_lev, double scale, unsigned flags,
                               const double* f, double* out);

/* Element blocks of the operators that take a second field (today: MIMSEM_OP_UMAT_RAY, f = exner_k, u = exner_s,
 * tau = dt) -- what MatAXPY(M1->M, 1.0, M1ray->M) needs (eul/Euler_2.cpp:1448).                  */
int mimsem_op_element_matrices_ex(mimsem_ctx* ctx, int op, int geom_lev, double scale, double tau, unsigned flags,
                                  const double* f, const double* u, double* out);

/* MatMult of caller-assembled element blocks: y_lev (+)= alpha * sum_e P_e^T B_e P_e x_lev, the blocks being what the
 * reference hands to MatSetValues(M, n, inds, n, inds, blk, ADD_VALUES) (eul/Assembly.cpp:128-131).  form 0/1/2 selects the
 * DoF lists (1-forms: x-edges then y-edges, block size 2*n1e).  blocks: device [nEl][nd][nd] row-major per level
 * (blocks_level_stride doubles apart; 0 = the same blocks on every level).  Used for element-block (PCBJACOBI-per-element,
 * eul/HorizSolve.cpp:77-96) preconditioners and for operators a caller modified entry-wise (MatAXPY).  flags: ACCUM, TRANSPOSE.
 * With blocks_level_stride = 0 the blocks stay in LDS while the kernel sweeps the levels, and elem_scale (nullable,
 * [nlev][elem_scale_stride >= nEl]) multiplies element e's result at level lev: B_e(lev) = elem_scale[lev][e] * B_e -- e.g. the
 * inverse 1-form mass blocks of all levels from ONE thickness-free inverse per element and 1/thickInv per (level, element). */
/* In-place inverse of nblocks dense n x n blocks (row-major, contiguous): the batched Gauss-Jordan behind WmatInv / WhmatInv and the
 * column operators (LinAlg::Inv, eul/LinAlg.cpp:186-269: natural-order sweep for the SPD mass blocks with the reference's full-pivoting
 * algorithm as the fall-back) for the element-block preconditioners of the Krylov solves (PCBJACOBI blocks, eul/HorizSolve.cpp:88-90):
 * one block per thread in LDS above n = 16; MIMSEM_ERR_UNSUPPORTED when a single block no longer fits 160 KB (n > ~140).                 */
int mimsem_block_inverse(mimsem_ctx* ctx, long long nblocks, int n, double* blocks);
/* The same with the count of blocks for which the reference's Inv reports an error (a pivot below 1e-12 under full pivoting,
 * eul/LinAlg.cpp:243-246; its callers ignore it, e.g. eul/VertOps.cpp:434): returns MIMSEM_OK and *n_singular >= 0; synchronises. */
int mimsem_block_inverse_status(mimsem_ctx* ctx, long long nblocks, int n, double* blocks, int* n_singular);
int mimsem_elem_blocks_apply(mimsem_ctx* ctx, int form, int nlev, unsigned flags, const double* blocks, long long blocks_level_stride,
                             const double* elem_scale, long long elem_scale_stride,
                             const double* x, long long x_stride, double* y, long long y_stride, double alpha);

/* Pvec::assemble / Phvec::assemble (Assembly.cpp:602-689): lumped 0-form mass as a vector */
int mimsem_pvec(mimsem_ctx* ctx, int geom_lev0, int nlev, double scale,
                const double* h2, long long h_stride, double* y, long long y_stride);

/* incidence matrices (E10mat :1102-1162, E21mat :1170-1220) applied as stencils.
 * which: 0 = E10 (0->1, own west/south edges), 1 = E21 (1->2), 2 = E12 = -E21^T (2->1), 3 = E01 = -E10^T (1->0) */
int mimsem_incidence_apply(mimsem_ctx* ctx, int which, int nlev,
                           const double* x, long long x_stride, double* y, long long y_stride);

/* Row A7, Geom::interp0 / interp1_l / interp2_l / interp1_g / interp2_g (eul/Geom.cpp:328-417) at EVERY quadrature point
 * of every element in one launch (the reference evaluates one point per call).  x: local (ghosted) k-form vectors, one row
 * per level.  out per level: form 0 or 2 -> [nEl][mp12]; form 1 -> [nEl][mp12][2] (u, v).  flags & MIMSEM_INTERP_GLOBAL
 * applies the Piola push-forward of the _g variants (1-forms: J/det; 2-forms: 1/det); form 0 ignores it. */
#define MIMSEM_INTERP_GLOBAL 1u
int mimsem_interp_quad(mimsem_ctx* ctx, int form, unsigned flags, int nlev,
                       const double* x, long long x_stride, double* out, long long out_stride);

/* Row N3: the packed [u,h] operator of the shallow-water Picard step, SWEqn::assemble_operator (src/SWEqn_Picard.cpp:622-725),
 * which the reference forms with MatMatMult / MatGetRow / MatSetValues and hands to KSPSolve(kspA):
 *     y_u = (M1 + a R(f)) u + a g E12 M2 h        y_h = M2 (a H E21 u + h)        a = ROS_ALPHA dt, g = grav, H = H_MEAN
 * applied matrix-free in ONE element pass plus the 1-form gather (all four blocks are element-local up to that gather).
 * x, y: rows of packed vectors [u (n1 slots) | h (n2 slots)]; f0: the Coriolis 0-form (SWEqn::fg), one row per level row
 * (f0_stride = 0 broadcasts one row).  Unit `scale`, thickness of geometry level 0 (the src/ flavour has neither). */
int mimsem_sw_operator_apply(mimsem_ctx* ctx, int nlev, double a, double grav, double H,
                             const double* f0, long long f0_stride,
                             const double* x, long long x_stride, double* y, long long y_stride);

/* Element-block preconditioner on packed [u | h] vectors (a PCSHELL for kspA, src/SWEqn_Picard.cpp:635-640; the reference uses
 * PETSc's default block-Jacobi/ILU(0) on the assembled A):  z = sum_e R_e^T B_e R_e r,  R_e = the element's 2 n1e edge slots
 * followed by its n2e face slots.  blocks: [nEl][ND][ND] with ND = 2 n1e + n2e, COLUMN-major per element (entry (r,c) at
 * [c*ND + r]).  Orders 1..4 (ND <= 64); higher orders return MIMSEM_ERR_UNSUPPORTED.  x != y. */
int mimsem_sw_blocks_apply(mimsem_ctx* ctx, int nlev, const double* blocks,
                           const double* x, long long x_stride, double* y, long long y_stride);

/* Richardson sweeps on the engine operators (row N1: the KSPSolve that follows almost every apply, eul/HorizSolve.cpp:77-96,
 * src/SWEqn_Picard.cpp:84-92, :348-353), for systems whose preconditioned operator is close to the identity.  The operator
 * result is never written: the gather pass applies the update  x += P (b - Op x)  directly.  upd (may be NULL) receives the
 * update P (b - Op x) itself -- its norm is the preconditioned residual a KSP monitors.
 *   mimsem_op_richardson_sweep:    P = diag(dinv); any operator with equal 0- or 1-form input and result spaces (incl. the
 *                                  upwinded ones: tau, u as in mimsem_op_apply_up); two launches.
 *   mimsem_block_richardson_sweep: P = sum_e R_e^T B_e R_e on 1-forms, blocks [nEl][2 n1e][2 n1e] COLUMN-major per element
 *                                  (the PCBJACOBI-per-element preconditioner with its inverse blocks supplied); three launches. */
int mimsem_op_richardson_sweep(mimsem_ctx* ctx, int op, int geom_lev0, int nlev, double scale, double tau, unsigned flags,
                               const double* f, long long f_stride, const double* u, long long u_stride,
                               const double* b, long long b_stride, const double* dinv, long long dinv_stride,
                               double* x, long long x_stride, double* upd, long long upd_stride);
/* mimsem_op_richardson_sweep with the Chebyshev update (round 5): z = dinv (b - Op x);  p = z + beta p;  x += alpha p.  With alpha, beta from
 * the region of the spectrum of diag(dinv) Op -- for the upwinded lumped 0-form mass of SWEqn::diagnose_q (src/SWEqn_Picard.cpp:322-341) a
 * vertical segment 1 +- 0.27 i, i.e. an ellipse with imaginary foci: 16 steps where plain sweeps take 25 -- a fixed-length solve. */
int mimsem_op_chebyshev_sweep(mimsem_ctx* ctx, int op, int geom_lev0, int nlev, double scale, double tau, unsigned flags,
                              const double* f, long long f_stride, const double* u, long long u_stride,
                              const double* b, long long b_stride, const double* dinv, long long dinv_stride, double alpha, double beta,
                              double* p, long long p_stride, double* x, long long x_stride, double* upd, long long upd_stride);
int mimsem_block_richardson_sweep(mimsem_ctx* ctx, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                                  const double* f, long long f_stride, const double* blocks,
                                  const double* b, long long b_stride, double* x, long long x_stride,
                                  double* upd, long long upd_stride);

/* One step of the Chebyshev semi-iteration for  Op x = b  on 1-forms with the element-block preconditioner
 * P = sum_e R_e^T (elem_scale[lev][e] B_e) R_e  (elem_scale may be NULL):   z = P (b - Op x);  p = z + beta p;  x += alpha p.
 * The caller supplies alpha, beta of the step (they depend only on the spectral bounds of P Op and the step number --
 * mimsem_amd/krylov.py ChebyshevMass) and a direction vector p that persists between steps.  No inner products, three
 * launches, nothing but x, p (and upd = z if given) written: a mass solve with a FIXED number of steps is hipGraph-capturable. */
int mimsem_block_chebyshev_sweep(mimsem_ctx* ctx, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                                 const double* f, long long f_stride, const double* blocks,
                                 const double* elem_scale, long long elem_scale_stride,
                                 const double* b, long long b_stride, double alpha, double beta, double* p, long long p_stride,
                                 double* x, long long x_stride, double* upd, long long upd_stride);

/* A whole FIXED-length Chebyshev solve of  Umat x = b  from x = 0 (what KSPSolve(ksp1, b, x) with a zero initial guess is in
 * eul/HorizSolve.cpp:224, 246, 305, 326): nsteps steps of mimsem_block_chebyshev_sweep with coef = {alpha_0, beta_0, alpha_1, beta_1, ...},
 * in ONE call: the first step needs no operator pass (Op 0 = 0) and writes x and its direction vector instead of updating them, so
 * neither is cleared or kept by the caller.  The same bits as the sequence of sweeps on x = 0.  x: result (need not
 * be initialised); pb (may be NULL): receives P b, the first step's preconditioned residual; upd (may be NULL): the last step's -- the two
 * vectors of a convergence check.  op must be MIMSEM_OP_UMAT (flags: MIMSEM_FLAG_VERT or 0; f unused), orders <= 5.  Capturable once the
 * context's workspaces have the size (a first call outside the capture). */
int mimsem_block_chebyshev_solve(mimsem_ctx* ctx, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                                 const double* f, long long f_stride, const double* blocks,
                                 const double* elem_scale, long long elem_scale_stride,
                                 const double* b, long long b_stride, int nsteps, const double* coef,
                                 double* x, long long x_stride, double* pb, long long pb_stride, double* upd, long long upd_stride);

/* z = P (A x) for the left-preconditioned Krylov iteration on the shallow-water operator: mimsem_sw_operator_apply followed by
 * mimsem_sw_blocks_apply in three launches instead of four -- the block pass reads the operator's element-local results through
 * the gather plan, the assembled A x is never written. */
/* One step of the Chebyshev semi-iteration on the preconditioned shallow-water operator B = P A (mimsem_sw_operator_precond_apply's P A) in
 * the SAME three launches, its vector algebra riding in the block pass (2-form rows) and the gather (1-form slots):
 *   x += d;   r -= P A d;   d = ca d + cb r        on the packed [u | h] rows; d is updated in place.
 * With the coefficients of the step from the spectral interval of P A (mimsem_krylov_chebyshev_update's comment) a solve of
 * KSPSolve(kspA, f, dx) (src/SWEqn_Picard.cpp:751-765) is ~30 of these calls and no inner product.  x, r, d distinct; orders 1..4. */
int mimsem_sw_operator_precond_chebyshev(mimsem_ctx* ctx, int nlev, double a, double grav, double H, const double* f0, long long f0_stride,
                                         const double* blocks, double ca, double cb, double* x, long long x_stride,
                                         double* r, long long r_stride, double* d, long long d_stride);
/* The same step in TWO launches (round 5): the 1-form part of a step's update -- the gather pass above -- rides in the element pass of the NEXT
 * step.  A chain of calls:  step2(pending = 0, ...) ; step2(pending = 1, pca, pcb = the previous call's ca, cb, ...) ; ... ; flush(last ca, cb).
 * Packed [u | h] rows as above.  The 2-form rows are finished by every call in rh, dh (always the same arrays).  The 1-form rows of r and d are
 * READ from (r_in, d_in) and, when an update is pending, WRITTEN to (r_out, d_out): two pairs of arrays the caller alternates (the other
 * elements of an edge slot read the old values in the same launch); without a pending update nothing is written to them and d_in supplies
 * the direction.  mimsem_sw_chebyshev_flush applies the update the chain still owes at its end, in place on the pair that holds the current
 * 1-form rows.  Same arithmetic, same order, same bits as mimsem_sw_operator_precond_chebyshev; orders 1..4. */
int mimsem_sw_chebyshev_step2(mimsem_ctx* ctx, int nlev, double a, double grav, double H, const double* f0, long long f0_stride, const double* blocks,
                              int pending, double pca, double pcb, double ca, double cb, double* x, long long x_stride,
                              const double* r_in, const double* d_in, double* r_out, double* d_out, double* rh, double* dh, long long v_stride);
int mimsem_sw_chebyshev_flush(mimsem_ctx* ctx, int nlev, double ca, double cb, double* x, long long x_stride, double* r, double* d, long long v_stride);
int mimsem_sw_operator_precond_apply(mimsem_ctx* ctx, int nlev, double a, double grav, double H,
                                     const double* f0, long long f0_stride, const double* blocks,
                                     const double* x, long long x_stride, double* z, long long z_stride);

/* The same body TOGETHER with the first classical Gram-Schmidt pass of the Arnoldi step it feeds (round 4; one level row):
 *   w = P (A x);  h[0..k) = V w;  w += alpha V^T h      (V: k rows of length n1 + n2, stride ldv; alpha = -1 orthogonalises)
 * in four launches instead of five: the 1-form gather of w rides in the dot pass.  Bit-identical to
 * mimsem_sw_operator_precond_apply + mimsem_krylov_orthogonalize (same sums, same order).  Stands where KSPSolve's GMRES calls
 * PCApply(MatMult(A, v)) and then KSPGMRESClassicalGramSchmidtOrthogonalization (src/SWEqn_Picard.cpp:751-765). */
int mimsem_sw_operator_precond_orthogonalize(mimsem_ctx* ctx, double a, double grav, double H, const double* f0, const double* blocks,
                                             const double* x, double* w, int k, const double* V, long long ldv, double alpha, double* h);

/* ---- vertical / column operators (rows C1..C9), eul/VertOps.h:45-72 ------------------------- */
enum mimsem_colop {
    MIMSEM_V_CONST = 0, MIMSEM_V_CONST_INV = 1, MIMSEM_V_CONST_RHO = 2, MIMSEM_V_CONST_RHO_INV = 3,
    MIMSEM_V_CONST_THETA = 4, MIMSEM_V_EOS_BLOCK = 5, MIMSEM_V_LINEAR = 6, MIMSEM_V_LINEAR_INV = 7,
    MIMSEM_V_LINEAR_RT = 8, MIMSEM_V_LINEAR_THETA = 9, MIMSEM_V_LINEAR_RHO2 = 10, MIMSEM_V_RAYLEIGH = 11,
    MIMSEM_V_LINCON = 12, MIMSEM_V_LINCON2 = 13, MIMSEM_V_CONLIN = 14, MIMSEM_V_CONLIN_W = 15,
    MIMSEM_V_CONLIN_RHODPI = 16,
    /* Strang / Held-Suarez rows, through the *_ex entry points (scalar parameter and/or horizontal velocity):          */
    MIMSEM_V_LINEAR_RAYLEIGH_INV = 17, /* AssembleLinearWithRayleighInv(ex,ey,dt_fric,A) :1380-1413  param = dt_fric       */
    MIMSEM_V_EOS_BLOCK_INV = 18,       /* Assemble_EOS_BlockInv(ex,ey,rt,theta,B)        :1049-1142  f1 = rt, f2 = theta|NULL */
    MIMSEM_V_LINEAR_RHO2_UP = 19,      /* AssembleLinearWithRho2_up(ex,ey,rho,A,dt,uhl)  :1415-1490  f1 = rho, param = dt, uh  */
    MIMSEM_V_LINCON2_UP = 20,          /* AssembleLinCon2_up(ex,ey,AB,dt,uhl)            :1492-1561  param = dt, uh           */
    MIMSEM_V_COUNT
};
/* L2Vecs::HorizToVert (dir=0) / VertToHoriz (dir=1), eul/L2Vecs.cpp:55-101:
 * vh[k*h_stride + e*n2e + i] <-> vz[e*nkv*n2e + k*n2e + i], nkv levels (nk, nk-1 or nk+1)        */
int mimsem_l2_transpose(mimsem_ctx* ctx, int dir, int nkv, double* vh, long long h_stride, double* vz);

/* Block structure of a column operator for ALL columns: the (block-diagonal or block-bidiagonal)
 * n2e x n2e blocks VertOps::Assemble*(ex,ey,[Vec..],Mat) would MatSetValues.  Fields are "vertical"
 * vectors vz[e][k*n2e+i] (L2Vecs::vz).  out: device [nEl][nblk][n2e][n2e], nblk from
 * mimsem_colop_nblocks (block order documented in DESIGN.md).                                     */
int mimsem_colop_nblocks(const mimsem_ctx* ctx, int colop);
int mimsem_colop_blocks(mimsem_ctx* ctx, int colop, unsigned flags,
                        const double* f1, const double* f2, double* out);
/* y_e = A_colop(e) x_e for every column (MatMult(vo->VX, x, y) after vo->AssembleX(ex,ey,..)),
 * transpose != 0 applies A^T (MatMultTranspose, eul/VertSolve.cpp:486)                           */
int mimsem_colop_apply(mimsem_ctx* ctx, int colop, unsigned flags, int transpose,
                       const double* f1, const double* f2, const double* x, double* y);
/* MatMult(M, x, y) with blocks obtained earlier from mimsem_colop_blocks(_ex) -- for the operators that depend on the geometry
 * only (AssembleConst, AssembleConstInv, AssembleLinear, AssembleLinearInv, AssembleRayleigh: eul/VertOps.cpp:184-1013), which the
 * reference re-assembles before every MatMult */
int mimsem_colop_apply_blocks(mimsem_ctx* ctx, int colop, int transpose, const double* blocks, const double* x, double* y);

/* EOS / log / exp column vectors (C7), eul/VertOps.cpp:732-787, :987-1047, :1204-1305.
 * which: 0 = Assemble_EOS_Residual(rt=a, exner=b) ; 1 = Assemble_EOS_RHS(rt=a, factor=p0, exponent=p1) ;
 *        2 = AssembleConstWithLogThetaPlusEta(theta=a, eta=b or NULL) ; 3 = AssembleConstWithRhoExpEta(rho=a, eta=b) */
int mimsem_column_eos(mimsem_ctx* ctx, int which, const double* a, const double* b,
                      double p0, double p1, double* out);

/* C6: theta diagnosis per column.  which: 0 = diagTheta_L2 (eul/VertSolve.cpp:322-352),
 * 1 = diagTheta2 (:289-319; theta has nk+1 levels).  rho, rt, theta: vertical vectors.            */
int mimsem_column_diag_theta(mimsem_ctx* ctx, int which, const double* rho, const double* rt, double* theta);

/* C5: VertSolve::solve_schur_column_eta for every column (eul/VertSolve.cpp:677-823): builds the
 * block-tridiagonal Helmholtz operator L_pi analytically from its block-(bi)diagonal factors, solves
 * it with a batched block-Thomas sweep, back-substitutes.  F_* are updated in place exactly as the
 * reference does (they are read afterwards, :1868-1894); d_* are outputs.  All vertical vectors:
 * theta/rho/eta/pi/F_rho/F_eta/F_pi/d_rho/d_eta/d_pi [nEl][nk*n2e], F_u/d_u [nEl][(nk-1)*n2e].      */
int mimsem_column_solve_schur_eta(mimsem_ctx* ctx, double dt,
        const double* theta, const double* rho, const double* eta, const double* pi,
        double* F_u, double* F_rho, double* F_eta, double* F_pi,
        double* d_u, double* d_rho, double* d_eta, double* d_pi);
/* Per-column status of the most recent mimsem_column_solve_schur_eta of this context.  The block-Thomas sweep of the default path
 * (orders 1..4) does not pivot ACROSS blocks (the reference's PCLU, eul/VertSolve.cpp:645-653, pivots over the whole band); it is
 * followed by up to four steps of iterative refinement, and a column stops when its correction is below 1e-10 of its solution.
 * column_status[e] (host, [nEl], may be NULL): 0 = converged, 1 = NOT converged within the allowed steps (conditioning beyond what the
 * unpivoted sweep + refinement resolves to 1e-10: judge d_* of that column by column_ratio), 2 = refinement was switched off,
 * 3 = re-solved by the pivoted fallback (mimsem_column_set_pivot_fallback below, on by default) AND verified: every norm finite, normwise
 *     backward error of the pivoted solve <= 1e-12; column_ratio: of its refinement step (the conditioning-limited forward indicator).
 *     A flagged column the fallback cannot verify (NaN / Inf data, a singular system) keeps status 1, its d and a ratio that says why.
 * 4 = flagged for its CONDITIONING only: the block sweep's own solution has the normwise backward error of a pivoted LU (<= 1e-14 in the
 *     Frobenius norm; measured by the fallback's triage before any re-solve) and stands unchanged; column_ratio stays above 1e-10 -- the
 *     forward error no solver in double precision can bring down at cond ~ 1e10 (the reference's PCLU included).  Not counted in *n_unconverged.
 * column_ratio[e] (host, [nEl], may be NULL): |last correction| / |solution| of the column -- how far its refinement got (on rough random
 * columns with cond(L) ~ 1e10 it settles near 1e-9, where LAPACK's pivoted LU leaves the same residual; a ratio >> 1e-8 is a failed solve).
 * *n_unconverged = number of columns with status 1, or -1 when the last solve ran on a path that keeps no status (orders >= 5,
 * MIMSEM_SCHUR_FUSED=rows|wave|0: pivoted Gauss-Jordan inside the blocks + one refinement step).  Synchronises the context's stream. */
int mimsem_column_solve_status(mimsem_ctx* ctx, int* n_unconverged, int* column_status, double* column_ratio);
/* The remedy for status 1 -- ON BY DEFAULT since round 5 (the reference solves every column with PCLU, eul/VertSolve.cpp:645-653, :783-789,
 * :806-812; MIMSEM_COLUMN_PIVOT_FALLBACK=0 at context creation or on == 0 here switch it off): every mimsem_column_solve_schur_eta / _3
 * re-solves the columns its unpivoted sweep flags INSIDE the call, before the back substitution reads the solution -- an LU with partial
 * pivoting over the whole band (one wavefront per flagged column on the block structure, csrc/column_pivot.inc) plus one refinement step.
 * The triage first measures the backward error of the solution the column already has: at a pivoted LU's level the column is accepted as
 * it is (status 4); otherwise it is re-solved (status 3).  Either way it no longer counts in *n_unconverged.  Cost: one launch whose wavefronts read the
 * unconverged counter and leave when nothing is flagged; tens of microseconds (order 3) when something is -- the flagged columns run
 * concurrently, any number of them.  Orders 1..4, up to 1 024 unknowns per column (nk n2e); beyond that columns are left as they are.
 * on == 2: EVERY column goes through the pivoted LU (validation mode: the reference's algorithm for all columns, at its price). */
int mimsem_column_set_pivot_fallback(mimsem_ctx* ctx, int on);
/* Test infrastructure of the fallback's work distribution (replaces nothing in the reference): the NEXT column solve of this context treats
 * the n listed columns (host array, 0 <= column < nEl) as flagged by its block sweep -- status 1, counted -- whatever their refinement said,
 * so that a test can put more columns in front of the fallback's wavefronts than rough data ever flags (tests/test_gpu_fullsize.py).
 * One-shot: the list is consumed by that solve.  n == 0 clears a pending list.                                                        */
int mimsem_column_flag_for_test(mimsem_ctx* ctx, const int* columns, int n);
/* the assembled block-tridiagonal L_pi itself: out [nEl][nk][3][n2e][n2e] (sub, diag, super)       */
int mimsem_column_helmholtz_blocks(mimsem_ctx* ctx, double dt,
        const double* theta, const double* rho, const double* eta, const double* pi, double* out);

/* ---- row C8 fused: the residual assembly and update of the vertical Newton loop ------------------------------- */
/* VertSolve::solve_schur_eta (eul/VertSolve.cpp:1721-1973) is a Newton loop around solve_schur_column_eta; per iteration and column the
 * reference runs assemble_residual_ec with diagnose_F_z / diagnose_Phi_z (:237-286, :432-502: ~12 VertOps assemblies, ~25 MatMult), the
 * EOS residual, the entropy residual and the entropy variables (:1806-1851), and after the solve the update and the EOS (:1858-1912).
 * The three entry points below do that for EVERY column in four launches (orders 1..4 since round 4; MIMSEM_ERR_UNSUPPORTED above).  All arrays are
 * "vertical" vectors [nEl][slots*n2e]: velz*, F_w, d_w, add_w on the nk-1 interfaces, theta2 / blend2 on nk+1, the rest on the nk levels.
 *
 * mimsem_column_newton_residual: theta = theta_l2_h, Pi = exner_h (the time-centred fields), velz / rho / rt at times i and j, zv from
 *   VertSolve::initGZ, the half-time rho_h / rt_h, exner_j.  add_w / add_rho / add_rt (nullable) are added times dt to F_w / F_rho / F_rt:
 *   the u dw/dx term (:1809), the horizontal transport tendencies (:1799, :1822-1826), the Held-Suarez forcing (:1831-1834).
 *   Out: the four right-hand sides of solve_schur_column_eta (F_w, F_rho, F_eta, F_exner), theta_h in W3 and eta_h (its theta, eta
 *   arguments), and k2i [nEl][(nk-1)*n2e] = F_z . (VA(theta) grad Pi) entry by entry (its sum / SCALE is VertSolve::k2i_z). */
int mimsem_column_newton_residual(mimsem_ctx* ctx, double dt, double rayleigh,
        const double* theta, const double* Pi, const double* velz_i, const double* velz_j, const double* rho_i, const double* rho_j,
        const double* zv, const double* rt_i, const double* rt_j, const double* rho_h, const double* rt_h, const double* exner_j,
        const double* add_w, const double* add_rho, const double* add_rt,
        double* F_w, double* F_rho, double* F_eta, double* F_exner, double* th_w3, double* eta, double* k2i);
/* after the solve (:1858-1912): eta_j from the iterate BEFORE the update, velz_j += d_w, rho_j += d_rho, exner_j += d_exner,
 * rt_j = VB^-1 W^T Q (rho_j exp(eta_j)); x_h = 0.5 x_i + 0.5 x_j; norm_squares [8][nEl][nk*n2e] = squares of (d_exner, exner_j, d_w, velz_j,
 * d_rho, rho_j, d_eta, eta_j): summed per column they give the ratios of VertSolve::MaxNorm (:228). */
int mimsem_column_newton_update(mimsem_ctx* ctx, const double* d_w, const double* d_rho, const double* d_eta, const double* d_exner,
        const double* velz_i, const double* rho_i, const double* rt_i, const double* exner_i,
        double* velz_j, double* rho_j, double* rt_j, double* exner_j,
        double* velz_h, double* rho_h, double* rt_h, double* exner_h, double* norm_squares);
/* VertSolve::MaxNorm (:228) for the four pairs of mimsem_column_newton_update's norm_squares: out4 (device) = max over the columns of
 * sqrt(sum d^2 / sum x^2) for (exner, w, rho, eta) -- the rank-local operand of the MPI_Allreduce(MAX) of :1915-1918.  ratio_ws: device
 * workspace of 4 nEl doubles (the per-column ratios stay there).  Two launches; a NaN ratio wins the maximum. */
int mimsem_column_max_norms(mimsem_ctx* ctx, const double* norm_squares, double* ratio_ws, double* out4);
/* diagTheta2 and / or diagTheta_L2 (:289-352) in one launch, optionally blended: out = wa * theta(rho, rt) + wb * blend
 * (the half-time averages theta_h = 0.5 theta_j + 0.5 theta_i of :1896-1912).  theta2 / thetaL: either may be null. */
int mimsem_column_diag_theta_blend(mimsem_ctx* ctx, const double* rho, const double* rt, double* theta2, const double* blend2,
                                   double* thetaL, const double* blendL, double wa, double wb);

/* ---- Strang / Held-Suarez column rows --------------------------------------------------------- */
/* mimsem_colop_blocks / mimsem_colop_apply with the extra arguments of the *_ex operators: param (dt_fric or dt)
 * and uh = the horizontal velocity as local 1-forms, one row per level ([nk][uh_stride], the reference's Vec* uhl). */
int mimsem_colop_blocks_ex(mimsem_ctx* ctx, int colop, unsigned flags, double param, const double* f1, const double* f2,
                           const double* uh, long long uh_stride, double* out);
int mimsem_colop_apply_ex(mimsem_ctx* ctx, int colop, unsigned flags, int transpose, double param,
                          const double* f1, const double* f2, const double* uh, long long uh_stride,
                          const double* x, double* y);
/* C1 VertOps::vertOps (eul/VertOps.cpp:134-182): MatMult with V10 (which 0: [nEl][(nk-1)*n2e] -> [nEl][nk*n2e]),
 * V01 = -V10^T (which 1: nk -> nk-1 slots) or V10_full (which 2: nk+1 -> nk slots); pure +-1 stencils, bit-exact */
int mimsem_column_incidence(mimsem_ctx* ctx, int which, const double* x, double* y);
/* VertSolve::diagTheta_up (eul/VertSolve.cpp:354-384) for every column: theta [nEl][(nk+1)*n2e]                */
int mimsem_column_diag_theta_up(mimsem_ctx* ctx, double dt, const double* rho, const double* rt,
                                const double* uh, long long uh_stride, double* theta);
/* VertOps::AssembleTempForcing_HS (eul/VertOps.cpp:1563-1633): lat = latitude of the quadrature points
 * [nEl][mp12] (Geom::s[elInds0_l][1]); theta on the nk+1 interfaces; out [nEl][nk*n2e]                          */
int mimsem_column_temp_forcing_hs(mimsem_ctx* ctx, const double* lat, const double* exner, const double* theta,
                                  const double* rho, double* out);
/* VertSolve::solve_schur_column_3 (eul/VertSolve.cpp:504-675, RAYLEIGH friction on) for every column: the Schur
 * complement L_rt_rt is block-PENTAdiagonal; it is assembled band by band, regrouped into 2x2 super-blocks and
 * solved by the batched block-Thomas sweep.  theta on nk+1 interfaces, velz/F_u/d_u on nk-1, the rest on nk levels.
 * F_* are updated in place as the reference does.  L_out (optional) [nEl][nk][5][n2e][n2e], block column = row-2+b. */
#define MIMSEM_SCHUR3_NO_RAYLEIGH 1u   /* M_u_inv = AssembleLinearInv (box/VertSolve.cpp:30: RAYLEIGH undefined)          */
#define MIMSEM_SCHUR3_BOX_Q       2u   /* Q_rt_rho keeps VBA(pressure gradient): the box twin never re-assembles VBA(velz) */
#define MIMSEM_SCHUR3_BOX (MIMSEM_SCHUR3_NO_RAYLEIGH | MIMSEM_SCHUR3_BOX_Q)   /* box/VertSolve.cpp:879-1058 (config 5)  */
int mimsem_column_solve_schur_3(mimsem_ctx* ctx, double dt, unsigned flags,
        const double* theta, const double* velz, const double* rho, const double* rt, const double* pi,
        double* F_u, double* F_rho, double* F_rt, double* F_pi,
        double* d_u, double* d_rho, double* d_rt, double* d_pi, double* L_out);

/* ---- dense building blocks of the device Krylov solvers (the KSPGMRES solves, eul/HorizSolve.cpp:77-96, src/SWEqn_Picard.cpp:600-606)
 * V: Krylov basis stored row-wise [k][ldv] (ldv >= n).  mdot: h[i] = <V_i, w>, i < k (one pass over V, deterministic two-stage
 * reduction).  maxpy: w += alpha * sum_i h[i] V_i.  Together: one classical Gram-Schmidt pass.                                   */
int mimsem_krylov_mdot(mimsem_ctx* ctx, int k, long long n, const double* V, long long ldv, const double* w, double* h);
int mimsem_krylov_maxpy(mimsem_ctx* ctx, int k, long long n, const double* V, long long ldv, const double* h, double alpha, double* w);
/* One classical Gram-Schmidt pass of the GMRES Arnoldi step in two launches: h = V w, then w += alpha V^T h (alpha = -1) */
int mimsem_krylov_orthogonalize(mimsem_ctx* ctx, int k, long long n, const double* V, long long ldv, double alpha, double* w, double* h);
/* End of the Arnoldi step: v = w/|w|, col[0..k) = h1 + h2 (h2 may be NULL), col[norm_slot] = |w|.  col may be pinned host
 * memory (hipHostMalloc): the Hessenberg column then reaches the host without a copy of its own. */
int mimsem_krylov_normalize(mimsem_ctx* ctx, long long n, const double* w, double* v, int k, const double* h1, const double* h2,
                            double* col, int norm_slot);
/* Second Gram-Schmidt pass and normalisation together: h2 = V w; w -= V^T h2; v = w/|w|, col[0..k) = h1 + h2, col[norm_slot] = |w|.
 * Two launches (round 3): the k dots of the pass and w.w in one, then update + normalisation + column in one, with
 * |w - V^T h2|^2 = w.w - h2.h2 (exact for an orthonormal V; h2 is the small correction of a RE-orthogonalisation, nothing cancels).
 * Should more than half of w.w sit in h2 -- the first pass lost its orthogonality altogether -- the word given to
 * mimsem_krylov_gs_control is set to 1 and the caller repeats the step with the three-launch form (norm accumulated from the updated
 * vector; fused = 0 selects it, also MIMSEM_GS_FUSED_NORM=0).  flag: device or pinned host memory, NULL = none. */
int mimsem_krylov_gs_control(mimsem_ctx* ctx, int fused /* 0 | 1, < 0: leave */, int* flag);
int mimsem_krylov_reorthonormalize(mimsem_ctx* ctx, int k, long long n, const double* V, long long ldv, double* w, double* v,
                                   const double* h1, double* h2, double* col, int norm_slot);
/* The same with the form and the flag word as ARGUMENTS (round 4; the context-wide word of mimsem_krylov_gs_control was shared by every
 * solver object on the context: two GraphedGMRES instances overwrote each other's word).  fused != 0: the two-launch form, which sets
 * *flag = 1 (flag: device or pinned host memory, may be NULL) when more than half of w.w went into h2; a zero or non-finite w.w leaves
 * the flag alone and writes v = 0 instead of inf.                                                                                  */
int mimsem_krylov_reorthonormalize_ex(mimsem_ctx* ctx, int k, long long n, const double* V, long long ldv, double* w, double* v,
                                      const double* h1, double* h2, double* col, int norm_slot, int fused, int* flag);
/* Both Gram-Schmidt passes, the normalisation and the Hessenberg column of ONE Arnoldi step in three launches (round 4): the update of
 * the first pass and the dots of the second share a kernel (an entry of w is final as soon as its own update is done).  h1, h2: device
 * [k] (outputs); norm and flag as in the two-launch re-orthonormalisation: a raised *flag means "repeat the step with
 * mimsem_krylov_orthogonalize + mimsem_krylov_reorthonormalize_ex(fused = 0)".                                                     */
int mimsem_krylov_cgs2(mimsem_ctx* ctx, int k, long long n, const double* V, long long ldv, double* w, double* v,
                       double* h1, double* h2, double* col, int norm_slot, int* flag);
/* Batched CG (one independent system per row = per level; the ksp1 solves of all levels at once).  The per-row scalars stay in
 * device memory, so an iteration needs no host synchronisation:  rowdot: out[i] = <A_i, B_i> (deterministic two-stage reduction);
 * cg_update: alpha_i = num[i]/den[i], x_i += alpha_i p_i, r_i -= alpha_i Ap_i;  cg_direction: p_i = z_i + (num[i]/den[i]) p_i.      */
int mimsem_krylov_rowdot(mimsem_ctx* ctx, int nrows, long long n, const double* A, long long lda, const double* B, long long ldb, double* out);
/* Level-wise field algebra between the operator calls of HorizSolve (eul/HorizSolve.cpp: VecAXPY / VecAYPX / VecPointwiseDivide /
 * VecScale chains, e.g. :246-254, :451-459, :472-493, :655-700): out[r][j] = alpha (a[r][j] op b[r][j]) + beta c[r][j] for nrows rows of n
 * entries at their own strides; op 0: a alone (b ignored), 1: a*b, 2: a/b; c may be null (no second term) and may alias out.
 * mimsem_interface_average: out[k] = (a[k-1] + a[k])/2 over nk levels from nk-1 interface rows, missing boundary interfaces left out. */
int mimsem_vec_combine(mimsem_ctx* ctx, int nrows, long long n, double alpha, const double* a, long long a_stride, int op,
                       const double* b, long long b_stride, double beta, const double* c, long long c_stride, double* out, long long out_stride);
int mimsem_interface_average(mimsem_ctx* ctx, int nk, long long n, const double* a, long long a_stride, double* out, long long out_stride);
int mimsem_krylov_cg_update(mimsem_ctx* ctx, int nrows, long long n, const double* num, const double* den,
                            const double* p, long long ldp, const double* Ap, long long ldap,
                            double* x, long long ldx, double* r, long long ldr);
/* The vector algebra of one step of a Chebyshev semi-iteration on a preconditioned operator B = P A whose spectrum is (close to) a real
 * interval -- KSPCHEBYSHEV in PETSc's terms; here the [u|h] system of SWEqn::solve (src/SWEqn_Picard.cpp:751-765: KSPSolve(kspA, f, dx)),
 * whose coupled element-block preconditioner leaves P A with Ritz values in [0.35, 1.18] and |Im| < 0.05 -- in ONE pass:
 * x += d;  r -= Bd;  d = a d + b r (the updated r), row by row.  No inner products: a and b follow from the spectral bounds alone. */
/* ... its start from x = 0 with the preconditioned right-hand side c = P b:  r = s c;  d = r / theta;  x = 0  (theta: the centre of the interval;
 * s = -1 when c = P f and the system is A dx = -f, as in SWEqn::solve; r may be c itself) -- one launch.                                  */
int mimsem_krylov_chebyshev_start(mimsem_ctx* ctx, int nrows, long long n, double s, double theta, const double* c, long long ldc,
                                  double* r, long long ldr, double* d, long long ldd, double* x, long long ldx);
/* The vector algebra of one Chebyshev step whose operator result y had to be completed over a halo first (sharded meshes; one context uses the
 * fused sweeps above):  z = dinv (b - y)  -- or z = y when dinv is NULL (y already is the preconditioned residual; b unused) --;  p = z + beta p;
 * x += alpha p;  upd = z if given.  Row-wise, one launch.                                                                                  */
int mimsem_krylov_chebyshev_px(mimsem_ctx* ctx, int nrows, long long n, double alpha, double beta, const double* y, long long ldy,
                               const double* b, long long ldb, const double* dinv, long long lddinv, double* p, long long ldp, double* x, long long ldx,
                               double* upd, long long ldupd);
/* ... and the end of the Picard iteration around it (src/SWEqn_Picard.cpp:757-765: VecAXPY(x, 1.0, dx); VecNorm(dx); VecNorm(x)):
 * x += dx;  out[0] = dx . dx;  out[1] = x . x (the updated x)  -- one launch; out: device, 2 doubles; the bits of the update followed by two
 * mimsem_krylov_rowdot calls.                                                                                                            */
int mimsem_krylov_axpy_dots(mimsem_ctx* ctx, long long n, const double* dx, double* x, double* out);
int mimsem_krylov_chebyshev_update(mimsem_ctx* ctx, int nrows, long long n, double a, double b, const double* Bd, long long ldBd,
                                   double* x, long long ldx, double* r, long long ldr, double* d, long long ldd);
int mimsem_krylov_cg_direction(mimsem_ctx* ctx, int nrows, long long n, const double* num, const double* den,
                               const double* z, long long ldz, double* p, long long ldp);

/* ---- rows N1..N3 from C / C++: the solve loops (round 4) ------------------------------------------------------------------------
 * What the reference does after almost every operator assembly (eul/HorizSolve.cpp:77-96: KSPCreate(ksp1); KSPSetOperators(ksp1, M1->M,
 * M1->M); KSPSetTolerances(ksp1, 1e-16, 1e-50, PETSC_DEFAULT, 1000); KSPSetType(KSPGMRES); PCSetType(PCBJACOBI);
 * PCBJacobiSetTotalBlocks(size*nElsX*nElsX); then KSPSolve(ksp1, b, x) at :224, :246, :310, :322; kspA of the shallow-water step,
 * src/SWEqn_Picard.cpp:600-606, :751-765) as ONE object of this library: operator = an engine operator applied matrix-free (all
 * `nlev` levels: independent systems for CG, one block-diagonal system for GMRES), the packed shallow-water operator, or a callback;
 * preconditioner = none, a diagonal, element blocks (the PCBJACOBI-per-element of the reference with exact block inverses), the
 * coupled [u|h] element blocks, or a callback.  Iterations run on the context's stream; the host reads one scalar set per
 * convergence test (CG: every `check_every` iterations; GMRES: the Hessenberg column through pinned memory each iteration).
 * All vectors are device pointers, rows `stride` doubles apart.                                                                   */
typedef struct mimsem_ksp mimsem_ksp;
enum mimsem_ksp_type { MIMSEM_KSP_CG = 0, MIMSEM_KSP_GMRES = 1 };
enum mimsem_ksp_reason {                        /* KSPConvergedReason, the values PETSc uses */
    MIMSEM_KSP_CONVERGED_RTOL = 2, MIMSEM_KSP_CONVERGED_ATOL = 3, MIMSEM_KSP_CONVERGED_ITS = 4,
    MIMSEM_KSP_DIVERGED_ITS = -3, MIMSEM_KSP_DIVERGED_BREAKDOWN = -5, MIMSEM_KSP_DIVERGED_NANORINF = -9 };
typedef int (*mimsem_ksp_apply_fn)(void* user, int nlev, const double* x, long long x_stride, double* y, long long y_stride);
int  mimsem_ksp_create(mimsem_ctx* ctx, int type, mimsem_ksp** out);                       /* KSPCreate + KSPSetType            */
void mimsem_ksp_destroy(mimsem_ksp* ksp);                                                  /* KSPDestroy                        */
/* KSPSetOperators: A = engine operator `op` on geometry levels [geom_lev0, geom_lev0 + nlev) with the arguments of mimsem_op_apply
 * (square operators: equal input and result space); f must stay valid until the last solve.                                        */
int  mimsem_ksp_set_operator(mimsem_ksp* ksp, int op, int geom_lev0, int nlev, double scale, unsigned flags,
                             const double* f, long long f_stride);
/* A = the packed [u|h] shallow-water operator of mimsem_sw_operator_apply (rows of n1 + n2 doubles)                                */
int  mimsem_ksp_set_operator_sw(mimsem_ksp* ksp, int nlev, double a, double grav, double H, const double* f0, long long f0_stride);
/* A = caller's routine (a MATSHELL): y = A x for `nlev` rows of `n` doubles, on the context's stream                               */
int  mimsem_ksp_set_operator_shell(mimsem_ksp* ksp, int nlev, long long n, mimsem_ksp_apply_fn fn, void* user);
/* PCSetType: */
int  mimsem_ksp_set_pc_none(mimsem_ksp* ksp);
int  mimsem_ksp_set_pc_jacobi(mimsem_ksp* ksp, const double* dinv, long long dinv_stride);              /* z = dinv .* r       */
/* PCBJACOBI with one block per element, built HERE from the operator given to mimsem_ksp_set_operator (1-form mass-like operators:
 * UMAT / UHMAT / UTMAT(_H)): P = sum_e R_e^T D_e (A_e)^-1 D_e R_e with A_e the element's dense 2 n1e x 2 n1e block at geometry
 * level geom_lev0 without its thickness factor, D_e = 1 / (number of elements sharing the edge), times 1 / mean(thickInv) of the
 * element per level when the operator carries the thickness flag (exact where a layer's thickness is uniform over an element).     */
int  mimsem_ksp_set_pc_bjacobi(mimsem_ksp* ksp);
/* caller-built element blocks, as mimsem_elem_blocks_apply takes them (form 0 / 1 / 2; elem_scale may be NULL)                      */
int  mimsem_ksp_set_pc_elem_blocks(mimsem_ksp* ksp, int form, const double* blocks, const double* elem_scale, long long elem_scale_stride);
/* coupled [u|h] element blocks of mimsem_sw_blocks_apply (with the shallow-water operator: the fused z = P A x of
 * mimsem_sw_operator_precond_apply is used inside the iteration)                                                                   */
int  mimsem_ksp_set_pc_sw_blocks(mimsem_ksp* ksp, const double* blocks);
/* the same blocks built HERE from the operator given to mimsem_ksp_set_operator_sw (what PCSetUp does with the matrix of KSPSetOperators):
 * A_e = [[M1_e + a R_e(f0), a g E12_e M2_e], [a H M2_e E21_e, M2_e]] per element, inverted, weighted by 1 / edge multiplicity; orders 1..4 */
int  mimsem_ksp_set_pc_sw_bjacobi(mimsem_ksp* ksp);
int  mimsem_ksp_set_pc_shell(mimsem_ksp* ksp, mimsem_ksp_apply_fn fn, void* user);                       /* PCSHELL             */
/* KSPSetTolerances (restart: GMRES only, PETSc's default 30; check_every: CG iterations between two looks at the residual, <= 0: 2) */
int  mimsem_ksp_set_tolerances(mimsem_ksp* ksp, double rtol, double atol, int maxit, int restart, int check_every);
int  mimsem_ksp_set_initial_guess_nonzero(mimsem_ksp* ksp, int flag);                                    /* KSPSetInitialGuessNonzero */
/* KSPSolve.  CG monitors |b - A x| / |b| per row (every row must meet rtol or atol); GMRES the PRECONDITIONED residual as PETSc's
 * default does.  Returns MIMSEM_OK also when the iteration limit was hit: ask mimsem_ksp_get_info.                                  */
int  mimsem_ksp_solve(mimsem_ksp* ksp, const double* b, long long b_stride, double* x, long long x_stride);
/* KSPGetIterationNumber / KSPGetResidualNorm / KSPGetConvergedReason of the last solve (rnorm: relative, worst row)                 */
int  mimsem_ksp_get_info(const mimsem_ksp* ksp, int* iterations, double* rnorm, int* reason);
/* Eigenvalues of a real upper Hessenberg matrix H [n][n] (host, row-major, entries below the first subdiagonal ignored; n <= 400) -- the
 * host-side step of a Ritz estimate (what KSPComputeEigenvalues gets from LAPACK's hseqr in PETSc): a host that runs its own Arnoldi
 * process (a SHARDED mesh, where the inner products are all-reduced by the host: mimsem_amd/host/mimsem_sweqn.hpp) hands the small
 * matrix here.  wr / wi [n]: real and imaginary parts, unordered.  Pure host arithmetic (csrc/hqr_host.hpp), no GPU call.              */
int  mimsem_hessenberg_eigenvalues(int n, const double* H, double* wr, double* wi);


/* Two INDEPENDENT fixed-length Chebyshev solves of one shallow-water Picard iteration, both from a ZERO initial guess, in shared launches
 * (round 6): the 1-form mass system of diagnose_F (src/SWEqn_Picard.cpp:253-284: nA steps of {element pass, block pass, gather epilogue}) and the
 * upwinded lumped 0-form mass system of diagnose_q (:322-341: nB steps of {element pass, gather epilogue}) read nothing of each other; at the
 * ~3 500 elements of the src/ drivers every launch is a ~5 us dispatch floor, and launch k of both chains goes out as ONE grid
 * (csrc/elem_kernels.hip: k_sw_pair -- the bodies of the very kernels the two sweep entry points use: the same bits).  Equivalent to, on
 * x1 = x0 = 0, nA calls of mimsem_block_chebyshev_sweep(ctx, MIMSEM_OP_UMAT, 0, 1, 1.0, 0, NULL, 0, blocks1, NULL, 0, b1, 0, coefA[2k], coefA[2k+1],
 * p1, 0, x1, 0, last ? upd1 : (first ? pb1 : NULL), 0) and nB calls of mimsem_op_chebyshev_sweep(ctx, MIMSEM_OP_PHMAT_UP, 0, 1, 1.0, tau, 0, h, 0, u, 0,
 * b0, 0, dinv, 0, coefB[2k], coefB[2k+1], p0, 0, x0, 0, last ? upd0 : (first ? pb0 : NULL), 0) -- without the first steps' operator passes (Op 0 = 0)
 * and without cleared vectors: x1, p1, x0, p0 are OUTPUTS / workspaces and need not be initialised.  pb1 / pb0 (may be NULL): the first steps'
 * preconditioned residuals, P b1 and dinv b0 -- the reference norms of the convergence checks; upd1 / upd0 (may be NULL): the last steps'.
 * coefA / coefB: HOST arrays of (alpha, beta) pairs; single level, scale 1, no flags (the src/ flavour); orders 2..4 (MIMSEM_ERR_UNSUPPORTED
 * otherwise: call the sweeps).  3 nA - 1 and 2 nB - 1 launches in max of the two grids.                                                   */
int mimsem_sw_dual_chebyshev(mimsem_ctx* ctx, int nA, const double* coefA, const double* blocks1, const double* b1, double* p1, double* x1, double* upd1, double* pb1,
                             int nB, const double* coefB, double tau, const double* h, const double* u, const double* b0, const double* dinv,
                             double* p0, double* x0, double* upd0, double* pb0);

/* ---- halo exchange plan (replaces VecScatter gtol_0/gtol_1, eul/Topo.cpp:145-155) ------------ */
/* Pack/unpack kernels only: the transport (RCCL send/recv over xGMI) is driven by the host layer
 * (torch.distributed / ncclSend-ncclRecv) on the buffers these calls fill.
 * idx: device int32 list of vector slots; buf: device [nlev][count].                              */
/* The same for ALL neighbour ranks in one launch: idx = concatenated slot lists, segment s = [seg_off[s], seg_off[s+1])
 * (host array, nseg+1 entries); buf is segment-major [segment][level][slot], i.e. one contiguous message per neighbour
 * (alltoallv layout).  mode 0 = pack (v -> buf), 1 = unpack/INSERT, 2 = unpack/ADD; only segments [seg_begin, seg_end) are
 * touched (callers split an ADD into ranges without repeated target slots to keep the sums ordered).                      */
#define MIMSEM_HALO_MAX_SEGMENTS 64
int mimsem_halo_segments(mimsem_ctx* ctx, const int* idx, int nseg, const int* seg_off, int seg_begin, int seg_end,
                         int nlev, int mode, double* buf, double* v, long long v_stride);
int mimsem_halo_pack(mimsem_ctx* ctx, const int* idx, int count, int nlev,
                     const double* v, long long v_stride, double* buf);
/* mode 0 = INSERT_VALUES (SCATTER_FORWARD ghost fill), 1 = ADD_VALUES (SCATTER_REVERSE reduce)     */
int mimsem_halo_unpack(mimsem_ctx* ctx, const int* idx, int count, int nlev, int mode,
                       const double* buf, double* v, long long v_stride);

/* ---- the whole exchange behind the ABI: VecScatterBegin / VecScatterEnd on gtol_0 / gtol_1 ---------------------------------
 * (eul/Topo.cpp:145-155; REVERSE/ADD call sites eul/Assembly.cpp:2194-2195, FORWARD/INSERT eul/Euler_2.cpp:1455-1456).
 * A plan holds, per neighbour rank i (at most MIMSEM_HALO_MAX_SEGMENTS), the local slots whose values are SENT to it
 * (send_idx[send_off[i] .. send_off[i+1])) and the local slots that RECEIVE from it; slots are positions in a [level][nslots]
 * vector of this context, all levels of a field travel in one message per neighbour ([level][slot] inside the message).
 * For the reference's two scatters build two plans: REVERSE/ADD sends the ghost slots and receives into the owned mirrors,
 * FORWARD/INSERT the other way round (mimsem_amd/partition.py::build_plans produces both lists).
 *   mimsem_halo_begin packs on the context's stream and starts the transport on the plan's own communication stream;
 *   work enqueued on the context's stream between begin and end (interior elements) overlaps the exchange;
 *   mimsem_halo_end makes the context's stream wait for the transport and unpacks (ADD: in ranges of neighbours with
 *   disjoint target slots, i.e. in a fixed order -- reproducible sums).  One exchange in flight per plan.
 * Transports (choose one after create): RCCL point-to-point over xGMI on the HOST's communicator (librccl is resolved at run
 * time, the process's already-loaded copy if there is one); a host callback (GPU-aware MPI, torch.distributed ...) that must
 * leave the data in `recv` in stream order of `stream` (or synchronise it itself); loop-back (every neighbour is the rank itself).
 * Errors: MIMSEM_ERR_ARG (bad lists / mode / level count), MIMSEM_ERR_STATE (no transport set, begin while in flight, end without
 * begin, RCCL not loadable or failing, callback returned non-zero).                                                              */
typedef struct mimsem_halo mimsem_halo;
#define MIMSEM_HALO_INSERT 0     /* INSERT_VALUES (SCATTER_FORWARD ghost fill) */
#define MIMSEM_HALO_ADD    1     /* ADD_VALUES (SCATTER_REVERSE reduce) */
/* send / recv: device buffers, neighbour i's message at doubles [send_off[i], send_off[i+1]) / [recv_off[i], recv_off[i+1]) */
typedef int (*mimsem_halo_transport_fn)(void* user, const double* send, const long long* send_off, double* recv,
                                        const long long* recv_off, int nneigh, const int* ranks, void* stream);
int  mimsem_halo_create(mimsem_ctx* ctx, int nneigh, const int* ranks, const int* send_idx, const int* send_off,
                        const int* recv_idx, const int* recv_off, int nslots, int max_nlev, mimsem_halo** out);
void mimsem_halo_destroy(mimsem_halo* plan);
int  mimsem_halo_set_rccl(mimsem_halo* plan, void* nccl_comm);        /* ncclComm_t of the host, ranks as in that communicator */
/* The RCCL entry points are taken from the library instance that created the communicator: the handle the host passes here (its
 * dlopen handle of librccl; call before the first mimsem_halo_set_rccl, MIMSEM_ERR_STATE afterwards), else the copy the process has
 * already loaded under the name librccl.so[.1].  This library never loads a copy of its own (mimsem_halo_set_rccl returns
 * MIMSEM_ERR_STATE when neither exists).                                                                                          */
int  mimsem_halo_use_rccl_library(void* dl_handle);
int  mimsem_halo_set_transport(mimsem_halo* plan, mimsem_halo_transport_fn fn, void* user);
int  mimsem_halo_set_loopback(mimsem_halo* plan);
/* ONE-SIDED transport (round 6; replaces the same VecScatterBegin/End, eul/Assembly.cpp:2194-2195, eul/Euler_2.cpp:1455-1456): the plan's
 * receive buffer is exported with hipIpcGetMemHandle and opened by the neighbour ranks; mimsem_halo_begin then packs every message STRAIGHT
 * INTO the neighbour's receive buffer (xGMI between the GPUs of a node) and publishes a per-exchange sequence number into its arrival flags,
 * mimsem_halo_end waits for the flags of all neighbours in a one-wavefront kernel and unpacks -- kernels only, no library call and no
 * communication stream on the critical path, recordable in a hipGraph (mimsem_graph_begin/_end) with the solver around it.  Set-up, once
 * per plan: every rank calls _peer_export, the host exchanges the blobs (MPI_Allgather of MIMSEM_HALO_PEER_BLOB bytes per rank and plan),
 * then every rank calls _set_peer with its neighbours' blobs in the plan's neighbour order.  Ranks of one node, one process per rank
 * (a rank may list itself: its own buffer).  The wait is bounded (~2 s): _peer_status reports an exchange that gave up.
 * Status: verified between processes on ONE GPU, bit-equal to the callback transport; not yet A/B-ed against RCCL between GPUs --
 * RCCL stays the default transport of the hosts.                                                                                       */
#define MIMSEM_HALO_PEER_BLOB 1024
int  mimsem_halo_peer_export(mimsem_halo* plan, int my_rank, void* blob /* MIMSEM_HALO_PEER_BLOB bytes, host */);
int  mimsem_halo_set_peer(mimsem_halo* plan, int my_rank, const void* neighbour_blobs /* nneigh x MIMSEM_HALO_PEER_BLOB bytes, host */);
int  mimsem_halo_peer_status(mimsem_halo* plan, unsigned long long* timed_out_seq);
int  mimsem_halo_begin(mimsem_halo* plan, int mode, int nlev, double* v, long long v_stride);
int  mimsem_halo_end(mimsem_halo* plan);

/* Self-test of the half-row block algebra of order 4 (csrc/column_dpp.inc, dpp::RowsH: a 16 x 16 block row split over two lanes, the layout the
 * order-4 walk of solve_schur_column_3 runs on): for ntask sets of blocks A, B [ntask][16][16] (row-major), vectors x [ntask][16] and
 * quadrature-point coefficients cq [ntask][25] (all device) it writes, per task, out[825] = { A B + B A (256), A^-1 by the unpivoted
 * Gauss-Jordan (256; A symmetric positive definite), W^T diag(cq) W (256), A x (16), sum_j W[q][j] x[j] at the 25 points, 16 probe values }.
 * Test infrastructure of the library itself (tests/test_gpu_rows_half.py compares with numpy); replaces nothing in the reference.
 * MIMSEM_ERR_UNSUPPORTED unless the context is of order 4.  Synchronises. */
int  mimsem_selftest_rows_half(mimsem_ctx* ctx, int ntask, const double* A, const double* B, const double* x, const double* cq, double* out);

/* Round 5: fixed-length solves from C / C++.  Under its element-block preconditioner the 1-form mass matrix, and under the coupled [u|h]
 * element blocks the shallow-water operator of SWEqn::solve (src/SWEqn_Picard.cpp:751-765), have a REAL, narrow spectrum; a Chebyshev
 * semi-iteration with a step count known in advance (mimsem_block_chebyshev_sweep, mimsem_sw_operator_precond_chebyshev,
 * mimsem_op_chebyshev_sweep) then replaces KSPSolve's GMRES -- no inner product, no host round trip, recordable in a hipGraph
 * (mimsem_graph_*).  What the host needs from the KSP object PCSetUp built:
 * mimsem_ksp_get_pc_blocks: the device blocks of mimsem_ksp_set_pc_bjacobi (1- or 2-form operators; *elem_scale: the per (level, element)
 *   factor, or NULL) or of mimsem_ksp_set_pc_sw_bjacobi / _sw_blocks (the coupled blocks, *elem_scale = NULL), owned by the KSP object;
 *   *nd = rows of a block.  MIMSEM_ERR_STATE for any other preconditioner.
 * mimsem_ksp_ritz: m Arnoldi steps of P A (operator and preconditioner of the object as set) on a fixed pseudo-random start vector and the
 *   eigenvalues of the m x m Hessenberg matrix (shifted QR on the host): the smallest / largest real part and the largest |imaginary part|
 *   of the Ritz values -- the interval (or ellipse) the Chebyshev coefficients are computed from.  Set-up cost: m operator applications. */
int  mimsem_ksp_get_pc_blocks(const mimsem_ksp* ksp, const double** blocks, const double** elem_scale, int* nd);
int  mimsem_ksp_ritz(mimsem_ksp* ksp, int m, double* re_min, double* re_max, double* im_max);

/* ---- hipGraph capture of a launch sequence, for hosts that carry no HIP toolchain (the C++ shim, a PETSc application) -------------------
 * The reference calls its operators one level at a time -- for (kk ...) { M1->assemble(kk, SCALE, true); MatMult(M1->M, x[kk], y[kk]); ... }
 * (eul/Euler_2.cpp:1427-1457, eul/HorizSolve.cpp:651-699) -- and on one rank's patch (144 elements) such a call is two ~4 us kernels
 * behind ~10 us of launch cost.  A host that keeps that loop can record it ONCE: mimsem_graph_begin puts the context's stream into
 * capture (a context on the default stream gets a stream of its own for the duration), every call of this library made on the context
 * until mimsem_graph_end is recorded instead of executed -- calls that synchronise, copy to pageable host memory or would have to grow a
 * workspace return MIMSEM_ERR_STATE (run the sequence once un-captured first) -- and mimsem_graph_launch replays the whole sequence with
 * one submission, in stream order with the context's other work: on the context's CURRENT stream (a later mimsem_ctx_set_stream /
 * _use_own_stream moves the replays with it; only a recording made on the capture stream lent to a default-stream context stays there).
 * A recording belongs to its context: after mimsem_ctx_destroy its launch returns MIMSEM_ERR_STATE (destroy it all the same).
 * Pointers are baked in: the replay reads and writes the same arrays (VecGetArray of the same Vecs).  Replaces: nothing in the reference (PETSc has no such facility); it is what "capture
 * launch-bound inner loops in hipGraphs" means for a host behind this C ABI.
 * Errors: MIMSEM_ERR_STATE (begin while capturing, end without begin, a capture invalidated by an illegal call), MIMSEM_ERR_HIP. */
typedef struct mimsem_graph mimsem_graph;
int  mimsem_graph_begin(mimsem_ctx* ctx);
int  mimsem_graph_end(mimsem_ctx* ctx, mimsem_graph** out);
int  mimsem_graph_launch(mimsem_graph* graph);
int  mimsem_graph_num_nodes(const mimsem_graph* graph);
void mimsem_graph_destroy(mimsem_graph* graph);

#ifdef __cplusplus
}
#endif
#endif
