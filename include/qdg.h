/*
 * qdg.h -- C ABI of the MI355X-native DG compressible-flow path for
 *          Quinoa/Inciter.
 *
 * Drop-in boundary: everything Inciter's `DG` chare reaches
 * through `inciter::DGPDE` for the CompFlow DG scheme, plus the two free
 * limiter functions and the SSP-RK3 update that `DG` runs itself.  Paths are
 * relative to the reference repository root.
 *
 *   entry point            replaces (reference interface)
 *   ---------------------  ------------------------------------------------
 *   qdg_ctx_create         dg::CompFlow ctor + the g_inputdeck values it reads
 *                          (src/PDE/CompFlow/DGCompFlow.hpp:80-93,
 *                           src/Control/Inciter/InputDeck/InputDeck.hpp:191-238)
 *   qdg_mesh_upload        the per-chare mesh arguments of every DGPDE call:
 *                          inpoel, coord, FaceData, geoFace, geoElem
 *                          (src/PDE/DGPDE.hpp:93-114, src/Inciter/FaceData.hpp:41-106)
 *   qdg_lhs                DGPDE::lhs        (DGPDE.hpp:89-90  -> Integrate/Mass.cpp:25-73)
 *   qdg_initialize         DGPDE::initialize (DGPDE.hpp:80-86  -> Integrate/Initialize.cpp:29-201)
 *   qdg_rhs                DGPDE::rhs        (DGPDE.hpp:93-104 -> DGCompFlow.hpp:130-195)
 *   qdg_dt                 DGPDE::dt         (DGPDE.hpp:107-114 -> DGCompFlow.hpp:206-406)
 *   qdg_limit              WENO_P1 / Superbee_P1 as called from DG::lim
 *                          (src/Inciter/DG.cpp:1251-1260, src/PDE/Limiter.cpp:29-316)
 *   qdg_state_* / qdg_stage / qdg_step
 *                          the resident form of DG::lim/dt/solve for one RK
 *                          stage / one time step (DG.cpp:1229-1282, 1360-1430,
 *                          1432-1508): fields stay in HBM between stages
 *   qdg_diag               ElemDiagnostics::compute_diag
 *                          (src/Inciter/ElemDiagnostics.cpp:116-215)
 *   qdg_halo_*             DG::next/comsol/lim/comlim/dt ghost plumbing
 *                          (DG.cpp:1009-1086, 1262-1282, 1315-1380)
 *   qdg_gen_* / qdg_bnd_faces
 *                          host-side mirrors of inciter::FaceData's ctor and of
 *                          the geometry generators (src/Inciter/FaceData.cpp:19-41,
 *                          src/Mesh/DerivedData.cpp:937-1491) and of the
 *                          boundary-face regeneration of the mesh loader
 *                          (src/Inciter/Partitioner.cpp:357-393)
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; it never throws.
 *    qdg_last_error() returns the message of the calling thread's last error.
 *  - host field arrays use the reference's `tk::Fields` (UnkEqComp) layout:
 *    U[e*nprop + c*rdof + k], R/L[e*(5*ndof) + c*ndof + k]
 *    (src/Base/Data.hpp:462-471), c in {rho, rho*u, rho*v, rho*w, rho*E}.
 *  - size_t arrays are the reference's std::vector<std::size_t>; int arrays its
 *    std::vector<int> (esuel, esuf; -1 = physical boundary).
 *  - the callee BORROWS every pointer for the duration of the call only.
 *  - a handle may be used from one thread at a time; no hidden global state.
 *  - all arithmetic is IEEE fp64 on the device.
 *  - there is NO CPU fallback: without a HIP device qdg_ctx_create fails.
 */
#ifndef QDG_H
#define QDG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qdg_ctx qdg_ctx;
typedef struct qdg_mesh qdg_mesh;

/* discr::flux (src/Control/Inciter/Options/Flux.hpp) */
enum { QDG_FLUX_HLLC = 0, QDG_FLUX_LAXFRIEDRICHS = 1,
       QDG_FLUX_UPWIND = 2 /* Transport only, Riemann/Upwind.hpp:35-55 */ };
/* discr::limiter (src/Control/Inciter/Options/Limiter.hpp) */
enum { QDG_LIMITER_NONE = 0, QDG_LIMITER_WENOP1 = 1, QDG_LIMITER_SUPERBEEP1 = 2 };
/* Problem policy (src/PDE/CompFlow/Problem.hpp:14-80) */
enum { QDG_PROBLEM_USER_DEFINED = 0, QDG_PROBLEM_SOD_SHOCKTUBE = 1,
       QDG_PROBLEM_SEDOV_BLASTWAVE = 2, QDG_PROBLEM_VORTICAL_FLOW = 3,
       QDG_PROBLEM_TAYLOR_GREEN = 4,
       /* Transport problem policies (src/PDE/Transport/Problem/SlotCyl.cpp:30-170) */
       QDG_PROBLEM_SLOT_CYL = 5,
       /* more CompFlow policies (RotatedSodShocktube.cpp:28-45, NLEnergyGrowth.cpp:28-190) */
       QDG_PROBLEM_ROTATED_SOD_SHOCKTUBE = 6, QDG_PROBLEM_NL_ENERGY_GROWTH = 7,
       /* more Transport policies (CylAdvect.cpp:28-129, GaussHump.cpp:28-125) */
       QDG_PROBLEM_CYL_ADVECT = 8, QDG_PROBLEM_GAUSS_HUMP = 9,
       /* CompFlow RayleighTaylor.cpp:28-175 (alpha, betax/y/z, p0, r0, kappa) */
       QDG_PROBLEM_RAYLEIGH_TAYLOR = 10,
       /* dg::Transport: TransportProblemShearDiff (src/PDE/Transport/Problem/ShearDiff.cpp:28-160) */
       QDG_PROBLEM_SHEAR_DIFF = 11 };
#define QDG_MAX_SCALARS 5   /* transported scalars of one dg::Transport system (diagnostics: 5 slots per kind) */
/* BC state functions (src/PDE/CompFlow/DGCompFlow.hpp:649-701) */
enum { QDG_BC_DIRICHLET = 1, QDG_BC_SYMMETRY = 2, QDG_BC_EXTRAPOLATE = 3,
       /* Transport only (src/PDE/Transport/DGTransport.hpp:163-168, 276-352) */
       QDG_BC_INLET = 4, QDG_BC_OUTLET = 5 };
/* which DGPDE: dg::CompFlow (5 conserved variables, DGCompFlow.hpp) or dg::Transport
 * with ncomp transported scalars (DGTransport.hpp:78-186; BASELINE config 1 has one).
 * Transport: flux UPWIND, problems SLOT_CYL / CYL_ADVECT / GAUSS_HUMP / SHEAR_DIFF, BCs Dirichlet/
 * Extrapolate/Inlet/Outlet, constant dt only (dg::Transport::dt returns max,
 * DGTransport.hpp:189-199), limiters as for CompFlow, p-adaptive DG with one scalar;
 * rows of every field are ncomp*ndof doubles, component-major (mark = c*rdof). */
enum { QDG_PDE_COMPFLOW = 0, QDG_PDE_TRANSPORT = 1 };

typedef struct qdg_config {
  int32_t struct_size;     /* = sizeof(qdg_config), ABI check */
  int32_t device;          /* HIP device ordinal */
  int32_t ndof, rdof;      /* 1/1 (dg), 4/4 (dgp1), 10/10 (dgp2) */
  int32_t flux;            /* QDG_FLUX_* */
  int32_t limiter;         /* QDG_LIMITER_* */
  int32_t problem;         /* QDG_PROBLEM_* */
  int32_t nbc;             /* number of (side set, type) pairs below */
  const int32_t* bc_sideset; /* [nbc] side set ids (param::compflow::bc*) */
  const int32_t* bc_type;    /* [nbc] QDG_BC_* */
  double gamma, pstiff, cv;  /* param::compflow::gamma|pstiff|cv */
  double cweight;            /* discr::cweight (WENO central weight) */
  double alpha, beta, p0;    /* vortical_flow parameters */
  double cfl;                /* discr::cfl; used when dt <= 0 */
  double dt;                 /* discr::dt; > 0 selects constant time step */
  int32_t pde;               /* QDG_PDE_* */
  int32_t pref;              /* pref::pref: p-adaptive DG (scheme pdg; needs ndof = rdof = 4,
                                Grammar.hpp:399-407): per-element ndof in {1,4} */
  double tolref;             /* pref::tolref (default 0.1, InputDeck.hpp:232) */
  double betax, betay, betaz, r0, ce, kappa;   /* nl_energy_growth parameters (with alpha) */
  int32_t ncomp;             /* dg::Transport: component::transport, 1..QDG_MAX_SCALARS (0 = 1); CompFlow: 0 or 5 */
  int32_t reserved0;
  /* shear_diff parameters per scalar (param::transport::u0 | lambda | diffusivity,
     ShearDiff.cpp:43-45): u0[ncomp], lambda[2*ncomp], diffusivity[3*ncomp]; null otherwise */
  const double* tr_u0;
  const double* tr_lambda;
  const double* tr_diffusivity;
} qdg_config;

/* flattened std::map<int, std::vector<std::size_t>> FaceData::m_bface */
typedef struct qdg_bface {
  size_t nset;
  const int32_t* set_id;    /* [nset] */
  const size_t* set_off;    /* [nset+1] offsets into face */
  const size_t* face;       /* boundary face ids */
} qdg_bface;

const char* qdg_last_error(void);
const char* qdg_version(void);

int qdg_ctx_create(const qdg_config* cfg, qdg_ctx** out);
int qdg_ctx_destroy(qdg_ctx* ctx);
/* Device buffers freed by the library (meshes closed, temporaries of a mesh build) are kept for the
 * next allocation: on this platform hipMalloc of VRAM the process has used before costs ~34 ms per GiB,
 * which a re-mesh (DG::resizePostAMR's new Discretization, src/Inciter/DG.cpp:1537-1612) would pay on
 * every buffer.  The cache holds at most 40 % of the device's memory and is returned to the driver when an
 * allocation fails or by this call (released_bytes may be NULL) -- not with the last context: the next one
 * would pay the driver again. */
int qdg_device_pool_trim(size_t* released_bytes);
/* Take one region of `bytes` from the driver now; every later device allocation of the library that fits is
 * carved out of it (and goes back into it) without reaching the driver.  A run that will re-mesh (amr: dtref,
 * src/Inciter/Refiner.cpp:403-408) calls this once at start-up, sized by its memory budget: on this platform
 * hipMalloc of memory that has been used before costs ~34 ms per GiB, and a re-mesh at 80.9 M tets allocates
 * ~100 GB in dozens of pieces.  The region is returned by qdg_device_pool_trim / the last context once nothing
 * lives in it.  qdg_device_memory: free and total bytes of the context's device, bytes reserved this way. */
int qdg_device_pool_reserve(qdg_ctx* ctx, size_t bytes);
int qdg_device_memory(qdg_ctx* ctx, size_t* free_bytes, size_t* total_bytes, size_t* reserved_bytes);
/* Device memory of the context's device from the same pool, for the buffers a host hands back in: the packed
 * rows of qdg_state_rows_get / _put and qdg_halo_pack / _unpack (the message buffers of DG::comsol / comlim,
 * src/Inciter/DG.cpp:1023-1086, on the device).  Memory of the runtime the library itself uses, which
 * matters in a process that holds a second copy of the HIP runtime.  qdg_device_free(ctx, NULL) is a no-op. */
int qdg_device_alloc(qdg_ctx* ctx, size_t bytes, void** out);
int qdg_device_free(qdg_ctx* ctx, void* ptr);
/* run all kernels of this context on an existing HIP stream (hipStream_t) */
int qdg_ctx_set_stream(qdg_ctx* ctx, void* hip_stream);
int qdg_ctx_synchronize(qdg_ctx* ctx);
/* Tuning / A-B switches, by name; nothing but QDG_UPLOAD_STATS (timings on stderr) is read from
 * the process environment.  Set before the meshes they concern are created.  Unknown name: -1.
 *   "p1_rhs"        0 (default) tile / face-task kernels: every in-tile face once, LDS atomics
 *                   (results reproducible to rounding); 1 element-centric kernel, bitwise
 *                   reproducible run to run
 *   "fused_update"  1 (default) stage-0 RK update fused with the Superbee limiter of stage 1
 *   "renumber"      1 (default) Morton order of the device rows; 0 keeps the caller's order
 *   "host_layout"   1: qdg_mesh_from_connectivity runs qdg_mesh_upload's host layout code (A/B)
 *   "halo_depth"    2: chunks built from now on put the tets within TWO faces of a ghost last in the device order
 *                   (default 1: within one face), so that the send rows of a two-layer halo plan
 *                   (qdg_halo_set_depth) are trailing rows and their packs fold into the producing kernels
 *   "limiter_write_all" 1: the Superbee kernel writes back every tile (default: tiles it leaves unchanged are not
 *                   written; results identical) -- the step rate of a flow developed everywhere, for measurement
 *   "graph_step"    1: qdg_step_comm replays its launches as a hipGraph (see qdg_step_graph_status)
 *   "orient_by_gid" 1 (default): meshes built with global tet ids (qdg_mesh_upload_gid, qdg_mesh_from_chunk_gid)
 *                   orient their faces by global id; 0: the chare-local rule of src/Inciter/DG.cpp:480-483
 *   "keep_connectivity" 1: meshes built on the device without ghosts keep their connectivity, coordinates, esuel
 *                   and boundary faces resident (~ 60 B per tet) for qdg_mesh_refine_uniform; default 0
 *   "keep_pool"     0 (default): qdg_ctx_destroy of the process's last context returns the device buffer
 *                   cache to the driver; 1: keeps it for the next context */
int qdg_ctx_set_option(qdg_ctx* ctx, const char* name, int value);
int qdg_ctx_get_option(qdg_ctx* ctx, const char* name, int* value);
/* Problem::solution at n points (DGPDE::analyticSolution, src/PDE/DGPDE.hpp:141-144;
 * also the Dirichlet state and the initial condition): out[i*ncomp + c], ncomp = 5
 * (CompFlow) or the number of scalars (Transport); evaluated by the device functors of the
 * context's problem */
int qdg_solution(qdg_ctx* ctx, size_t n, const double* x, const double* y, const double* z,
                 double t, double* out);

/* Upload one mesh chunk.  nielem interior tets, nunk >= nielem incl. ghosts
 * (rows [nielem,nunk) of every field are ghosts filled by the halo exchange);
 * inpoel has 4*nunk entries, esuel 4*nielem, esuf 2*nfac, inpofa 3*nfac,
 * geoFace 7*nfac, geoElem 4*nunk; faces [0,nbfac) are physical-boundary
 * faces.  Interior elements are renumbered on the device for locality; all
 * host-facing arrays stay in the caller's numbering. */
int qdg_mesh_upload(qdg_ctx* ctx, size_t nielem, size_t nunk, size_t nnode,
                    const size_t* inpoel, const double* x, const double* y,
                    const double* z, size_t nbfac, size_t nfac,
                    const int* esuf, const int* esuel, const size_t* inpofa,
                    const double* geoFace, const double* geoElem,
                    const qdg_bface* bface, qdg_mesh** out);
/* The same with the GLOBAL ids of the chunk's tets (elem_gid[nunk], the numbering of the whole mesh
 * a serial run would see; NULL = qdg_mesh_upload).  With context option "orient_by_gid" (default 1) the
 * device mesh stores every interior and chare-boundary face with the tet of LOWER GLOBAL id as its left
 * tet -- the serial run's rule (src/Mesh/DerivedData.cpp:1127-1139: a face is kept by its lower-numbered
 * tet) -- instead of the chare-local one (left = the owned tet, src/Inciter/DG.cpp:480-483).  HLLC falls
 * through to the STORED right state where a wave speed is NaN (src/PDE/Integrate/Riemann/HLLC.hpp:93-124:
 * negative pressure at a face point next to a strong shock), so only then does a partitioned run take
 * the serial run's branches at such faces and equal it on every tet.  The caller's esuf / geoFace stay
 * as they are (left = owned); the flip happens in the device copy. */
int qdg_mesh_upload_gid(qdg_ctx* ctx, size_t nielem, size_t nunk, size_t nnode,
                        const size_t* inpoel, const double* x, const double* y,
                        const double* z, size_t nbfac, size_t nfac,
                        const int* esuf, const int* esuel, const size_t* inpofa,
                        const double* geoFace, const double* geoElem,
                        const qdg_bface* bface, const size_t* elem_gid, qdg_mesh** out);
int qdg_mesh_destroy(qdg_mesh* mesh);

/* -- stateless operators on host fields (the DGPDE-shaped calls) ---------- */
int qdg_lhs(qdg_mesh* mesh, double* L_aos);                       /* nunk rows */
int qdg_initialize(qdg_mesh* mesh, double t, double* U_aos);      /* rows [0,nielem) */
int qdg_rhs(qdg_mesh* mesh, double t, const double* U_aos, double* R_aos);
int qdg_dt(qdg_mesh* mesh, const double* U_aos, double* mindt);   /* unscaled min(vol/delt) */
int qdg_limit(qdg_mesh* mesh, double* U_aos);                     /* in place */

/* -- device-resident fast path ------------------------------------------- */
int qdg_state_upload(qdg_mesh* mesh, const double* U_aos);        /* nunk rows */
int qdg_state_download(qdg_mesh* mesh, double* U_aos);            /* nunk rows */
int qdg_state_initialize(qdg_mesh* mesh, double t);               /* IC projection on device */
/* phases of one RK stage, in the reference's order; the caller exchanges
 * ghosts (qdg_halo_*) between them exactly where DG::comsol / DG::comlim sit */
int qdg_stage_limit(qdg_mesh* mesh);
int qdg_stage_dt(qdg_mesh* mesh, double tleft); /* stage 0: local dt (CFL or constant, capped
                                                  to tleft) -> device scalar */
int qdg_stage_dt_get(qdg_mesh* mesh, double* dt_host);  /* sync + read local dt */
int qdg_stage_dt_set(qdg_mesh* mesh, double dt);        /* reduced dt back */
int qdg_stage_dt_device_ptr(qdg_mesh* mesh, void** dptr); /* for an in-place device min-reduction */
int qdg_stage_rhs_update(qdg_mesh* mesh, int stage, double t);
/* fused form used by qdg_step: the RHS of stage 0 also produces the local time
 * step (CFL sum taken from the Riemann solver's wave speeds; constant dt when
 * configured) into the dt scalar, capped to tleft; the caller min-reduces the
 * scalar across chunks, then calls qdg_stage_update.  Stages 1,2: RHS only. */
int qdg_stage_rhs_dt(qdg_mesh* mesh, int stage, double t, double tleft);
int qdg_stage_update(qdg_mesh* mesh, int stage);
/* p-adaptive DG (cfg.pref): the per-element number of DOFs DG::m_ndof lives on
 * the device (all = ndof after upload / qdg_state_initialize).  qdg_stage_pdg is
 * the stage-0 work of DG::next / DG::lim / DG::solve on it: eval_ndof
 * (DG.cpp:1088-1163), propagate_ndof (:1284-1313), zeroing of the high-order DOFs
 * of P0 elements (:1451-1469); call it before qdg_stage_limit of stage 0
 * (qdg_step does).  get/set: nunk entries in the caller's element numbering. */
int qdg_stage_pdg(qdg_mesh* mesh);
/* partitioned mesh: eval (DG::next) -> exchange ghosts (the slab rows carry the
 * tets' ndof as one more column, like comsol/comlim DG.cpp:1032,1275) ->
 * propagate + zeroing (DG::lim, DG::solve) */
int qdg_stage_pdg_eval(qdg_mesh* mesh);
int qdg_stage_pdg_propagate(qdg_mesh* mesh);
int qdg_ndofel_get(qdg_mesh* mesh, size_t* ndofel);
int qdg_ndofel_set(qdg_mesh* mesh, const size_t* ndofel);
/* whole step on one chunk without ghosts: 3 x (limit, [dt], rhs, update);
 * returns the dt taken (host sync only when dt_taken != NULL) */
int qdg_step(qdg_mesh* mesh, double t, double tleft, double* dt_taken);
/* sum_e sum_g wt*u^2, wt*(u-s)^2, max|u-s| per component (15 doubles) */
int qdg_diag(qdg_mesh* mesh, double t_new, double* out15);
/* Problem::fieldOutput on the device (as dg::CompFlow::fieldOutput / dg::Transport::fieldOutput
 * call it, DGCompFlow.hpp:447-462, DGTransport.hpp:248-279): EVERY field of the Problem's
 * own list -- numerical, analytical and err(.) fields -- from the cell means of the
 * resident state at time t (SodShocktube.cpp:139-258, VorticalFlow.cpp:131-254,
 * TaylorGreen.cpp:108-240, NLEnergyGrowth.cpp:208-318, RayleighTaylor.cpp:194-314,
 * UserDefined.cpp:87-169; the err(.) fields are x/V with V = 0 exactly as the reference
 * writes them); with p-adaptive DG the per-element ndof follows as the last field
 * (DG::writeFields, DG.cpp:1201-1204).  out[f*nielem + e], e in the caller's element
 * numbering, f < qdg_field_count; names from qdg_field_name (= Problem::fieldNames). */
int qdg_field_count(qdg_mesh* mesh, size_t* nfield);
const char* qdg_field_name(qdg_mesh* mesh, size_t f);
int qdg_field_output(qdg_mesh* mesh, double t, double* out);
/* the same without a mesh handle -- the stateless DGPDE members:
 *   qdg_ctx_field_count/name   DGPDE::fieldNames   (src/PDE/DGPDE.hpp:120-121)
 *   qdg_field_output_from      DGPDE::fieldOutput  (DGPDE.hpp:126-131): nunk rows of U and
 *                              geoElem in the caller's numbering, out[f*nunk + e] (no ndof field)
 *   qdg_avg_elem_to_node       DGPDE::avgElemToNode (DGPDE.hpp:134-139 ->
 *                              dg::CompFlow::avgElemToNode, DGCompFlow.hpp:465-552):
 *                              out[f*nnode + n], f = density, x/y/z velocity, specific total
 *                              energy, pressure averaged over the elements around node n
 *   qdg_initialize_from        DGPDE::initialize (DGPDE.hpp:80-86) before the chare's mesh is
 *                              uploaded -- DG::setup calls it first (DG.cpp:995-999) and its
 *                              arguments carry no FaceData: needs inpoel, coord and L only
 *                              (L may be NULL: volumes are then computed from coord);
 *                              writes rows [0,nielem) of U_aos */
int qdg_ctx_field_count(qdg_ctx* ctx, size_t* nfield);
const char* qdg_ctx_field_name(qdg_ctx* ctx, size_t f);
int qdg_field_output_from(qdg_ctx* ctx, double t, size_t nunk, const double* geoElem,
                          const double* U_aos, double* out);
int qdg_avg_elem_to_node(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                         const double* U_aos, double* out);
int qdg_initialize_from(qdg_ctx* ctx, size_t nielem, size_t nnode, const size_t* inpoel,
                        const double* x, const double* y, const double* z,
                        const double* L_aos, double t, double* U_aos);
/* WENO_P1 / Superbee_P1 with the reference's free-function inputs (Limiter.cpp:29-44,
 * 155-175: esuel, per-element ndofel, the solution) -- DG::lim runs before the first rhs/dt
 * of a run (DG.cpp:1251-1260), i.e. possibly before the chare's mesh is uploaded.
 * esuel has 4*nielem entries, U_aos nunk rows (in place); ndofel (nunk entries) may be NULL
 * unless the context is p-adaptive */
int qdg_limit_from(qdg_ctx* ctx, size_t nielem, size_t nunk, const int* esuel,
                   const size_t* ndofel, double* U_aos);
/* device pointer/stride of the resident SoA state, for zero-copy plumbing */
int qdg_state_device_ptr(qdg_mesh* mesh, void** dptr, size_t* stride);

/* -- ghost halo ---------------------------------------------------------- */
/* nnbr neighbours; send_elem lists local interior tets per neighbour (in the
 * order the receiver stores its ghosts), recv ranges are contiguous ghost
 * rows.  Buffers are device memory owned by the mesh handle, one contiguous
 * slab per direction, element-major rows of nprop doubles (like the
 * reference's comsol payload u[j] = m_u[tet]); with p-adaptive DG (cfg.pref)
 * nprop + 1 doubles, the last one the tet's ndof; qdg_halo_buffers reports
 * the row size. */
int qdg_halo_setup(qdg_mesh* mesh, size_t nnbr, const int32_t* nbr_rank,
                   const size_t* send_off, const size_t* send_elem,
                   const size_t* recv_off);
/* Two ghost layers (chunks of qdg_chunk_build_depth(depth = 2) built by qdg_mesh_from_chunk[_gid]; after
 * qdg_halo_setup with the plan's (rank, layer) entries): the first nghost1 ghost rows are layer 1.  From then on the
 * limiter stages (qdg_stage_limit, qdg_step_comm) also limit those rows -- layer 2 completes their inputs -- and
 * qdg_step_comm drops the exchange of the limited solution (DG::lim -> comlim, src/Inciter/DG.cpp:1262-1282):
 * 3 exchanges + 1 all-reduce per SSP-RK3 step instead of 6 + 1.  A driver that calls the stages itself skips its
 * second exchange of every stage.  Results equal the one-layer run's (the same limiter on the same inputs).
 * nghost1 = 0 returns to one layer.  Not for p-adaptive runs, nor for meshes of qdg_mesh_upload. */
int qdg_halo_set_depth(qdg_mesh* mesh, size_t nghost1);
/* the plan in use: its entries, the layer-1 ghosts the rank limits itself (0: one layer) and whether qdg_step_comm
 * can fold the halo packs into the kernels that produce the rows (every send row among the trailing device rows:
 * chunks built under context option "halo_depth" = 2 have that shape for two-layer plans); any pointer may be NULL */
int qdg_halo_info(qdg_mesh* mesh, size_t* nentry, size_t* nghost1, int32_t* packs_folded);
int qdg_halo_buffers(qdg_mesh* mesh, void** send_dev, void** recv_dev,
                     size_t* row_bytes);
/* use caller-owned device memory for the slabs / the dt scalar (e.g. buffers a
 * communication library registered); sizes as reported by qdg_halo_sizes */
int qdg_halo_use_buffers(qdg_mesh* mesh, void* send_dev, void* recv_dev);
int qdg_halo_sizes(qdg_mesh* mesh, size_t* nsend_rows, size_t* nrecv_rows);
int qdg_stage_dt_use_buffer(qdg_mesh* mesh, void* dt_dev);
/* local transport between chunks of ONE context (same device, same stream): send-slab rows of
 * src -> receive-slab rows of dst */
int qdg_halo_copy(qdg_mesh* dst, size_t dst_row0, qdg_mesh* src, size_t src_row0, size_t nrows);
int qdg_halo_pack(qdg_mesh* mesh);     /* U[send list] -> send slab */
int qdg_halo_unpack(qdg_mesh* mesh);   /* recv slab -> ghost rows of U */

/* -- RCCL transport (one process per GPU, xGMI) ----------------------------
 * Replaces the Charm++ messages of DG::next/comsol/lim/comlim (DG.cpp:1009-1086,
 * 1229-1282) by grouped ncclSend/ncclRecv between the ranks of halo_setup's
 * nbr_rank[], and contribute(min) of the time step (DG.cpp:1428-1429) by an
 * in-place ncclAllReduce(min) on the device scalar; everything is enqueued on
 * the context's stream, nothing synchronises the host.  RCCL is loaded with
 * dlopen at the first qdg_comm_* call (librccl.so.1).
 * Bootstrap: rank 0 calls qdg_comm_unique_id, the 128 bytes reach the other
 * ranks by any means (MPI, torch.distributed, a file), then every rank calls
 * qdg_comm_create at the same time (it is a collective). */
typedef struct qdg_comm qdg_comm;
int qdg_comm_unique_id(void* id128);
int qdg_comm_create(qdg_ctx* ctx, int nranks, int rank, const void* id128, qdg_comm** out);
int qdg_comm_destroy(qdg_comm* comm);
/* what RCCL itself reports for the communicator (ncclCommCount, ncclCommUserRank, ncclCommCuDevice):
 * the number of ranks that joined, this rank, its HIP device -- so that a run can state which ranks
 * its halo really crossed (any pointer may be NULL) */
int qdg_comm_info(qdg_comm* comm, int* nranks, int* rank, int* device);
int qdg_halo_exchange(qdg_mesh* mesh, qdg_comm* comm);   /* pack, send / recv into ghost rows */
int qdg_stage_dt_allreduce(qdg_mesh* mesh, qdg_comm* comm);
/* whole SSP-RK3 step of one chunk of a partitioned mesh:
 * 3 x (exchange, limit, exchange, rhs [+dt, min over ranks], update); the dt
 * taken is read back (host sync) only when dt_taken != NULL */
int qdg_step_comm(qdg_mesh* mesh, qdg_comm* comm, double t, double tleft, double* dt_taken);
/* Context option "graph_step" = 1: qdg_step_comm records its launch sequence (compute kernels, RCCL send /
 * receive / all-reduce kernels, ghost-row copy) once per buffer-rotation phase as a hipGraph and replays it with
 * one hipGraphLaunch per time step (the reference's analogue: nothing -- Charm++ schedules entry methods one
 * message at a time, src/Inciter/dg.ci:57-70).  Used where the step's kernels do not read t (Sod, Sedov, rotated
 * Sod), for uniform-order runs, outside profiling; the first two steps of a mesh run plain (RCCL connects its
 * peers lazily).  A graph is reused for calls with the SAME tleft (it is a kernel argument of the recorded dt
 * reduction): pass a constant (e.g. 1e300) while the end time is more than a step away and the true remainder only for
 * the last steps -- those run as plain launches (at most 8 graphs are kept per mesh).  Where HIP or RCCL refuses the capture the plain launches stay in charge.  Status: state 0 not
 * tried yet, 1 graphs in use, -1 refused (error text copied to `error`, any pointer may be NULL). */
int qdg_step_graph_status(qdg_mesh* mesh, int32_t* state, int32_t* ngraphs, int64_t* nreplays, char* error,
                          size_t error_len);

/* -- measurement ---------------------------------------------------------- */
/* When enabled, every launch of the RHS kernel inside qdg_stage_rhs_update is
 * bracketed by HIP events on the context's stream (no host sync in the timed
 * path); qdg_profile_read synchronises and returns the number of launches and
 * their summed duration since the last read. */
int qdg_profile_enable(qdg_mesh* mesh, int on);
int qdg_profile_read(qdg_mesh* mesh, size_t* nlaunch, double* total_ms);
/* The same events by kind -- [0] RHS launches, [1] halo exchanges (pack kernel + grouped ncclSend / ncclRecv, with
 * the gaps in front of them: DG::comsol / comlim, src/Inciter/DG.cpp:1023-1086, 1266-1358), [2] dt all-reduces
 * (contribute(min), DG.cpp:1428-1429) -- recorded inside qdg_step_comm, qdg_halo_exchange and
 * qdg_stage_dt_allreduce while profiling is enabled: counts and summed milliseconds since the last read. */
int qdg_profile_read_all(qdg_mesh* mesh, size_t count[3], double total_ms[3]);
/* algorithmic bytes of one RHS launch: nielem * (16*5*ndof + 32) + 24*nnode
 * (read U once, write R once, 8 int32 indices per tet, node coordinates once;
 * SURVEY.md 8d) */
int qdg_rhs_algorithmic_bytes(qdg_mesh* mesh, double* bytes);
/* the face-task mix of the mesh's tile layout (DG-P1 tile kernel): counts[0] interior faces with both tets in
 * one tile (evaluated once), [1] faces from a tile to another tile or to a ghost (evaluated by both sides),
 * [2] physical-boundary faces, [3] tiles -- the figure that says how a mesh's numbering suits the kernel
 * (what QDG_UPLOAD_STATS prints at build time) */
int qdg_mesh_layout_stats(qdg_mesh* mesh, size_t counts[4]);

/* -- host-side mesh-derived data (mirror of FaceData / DerivedData) ------- */
int qdg_gen_esuel(size_t nelem, const size_t* inpoel, int* esuel);
size_t qdg_gen_nipfac(size_t nelem, size_t nbfac, const int* esuel);
int qdg_gen_inpofa(size_t nelem, size_t nbfac, const size_t* inpoel,
                   const size_t* triinpoel, const int* esuel, size_t* inpofa);
int qdg_gen_belem(size_t nelem, size_t nbfac, const size_t* inpoel,
                  const size_t* inpofa, size_t* belem);
int qdg_gen_esuf(size_t nelem, size_t nbfac, const size_t* belem,
                 const int* esuel, int* esuf);
int qdg_gen_geoface(size_t nfac, const size_t* inpofa, const double* x,
                    const double* y, const double* z, double* geoFace);
int qdg_gen_geoelem(size_t nelem, const size_t* inpoel, const double* x,
                    const double* y, const double* z, double* geoElem);
/* boundary-face regeneration: ntri side-set triangles (3 node ids each, any
 * order) tagged with tri_set[]; out arrays sized >= ntri.  Faces are returned
 * grouped by ascending side set id, within a set in tet order. */
int qdg_bnd_faces(size_t nelem, const size_t* inpoel, size_t ntri,
                  const size_t* tri, const int32_t* tri_set, size_t* nbfac,
                  size_t* triinpoel, int32_t* face_set);

/* -- decomposition of an arbitrary tet mesh over the ranks of a run ----------
 * qdg_partition: geometric cut on the element centroids -- what Partitioner::partition
 * asks Zoltan2 for (src/Inciter/Partitioner.cpp:137-170: RCB, RIB, HSFC, MJ).  part[e] in
 * [0, nparts).  Both methods depend on the mesh alone (ties broken by element id).
 * qdg_chunk_build: one rank's chunk as the DG chare holds it after its ghost set-up
 * (src/Inciter/DG.cpp:134-949): the owned tets in input order, then one layer of ghost tets --
 * the tets of other ranks that share a face with an owned tet (DG.cpp:468-712) -- grouped by
 * owner rank (ascending) and ordered by global id inside a group; local node ids by first
 * touch; per neighbour the list of owned tets it needs, in the order in which it stores them
 * as ghosts.  esuel (4*nelem, FaceData::Esuel of the WHOLE mesh) may be NULL: it is then
 * generated (qdg_gen_esuel).  Feed the result to qdg_mesh_upload (after the FaceData / geometry
 * of the chunk) and qdg_halo_setup. */
enum { QDG_PART_RCB = 0, QDG_PART_MORTON = 1 };
typedef struct qdg_chunk qdg_chunk;
int qdg_partition(size_t nelem, const size_t* inpoel, size_t nnode, const double* x, const double* y,
                  const double* z, int nparts, int method, int32_t* part);
int qdg_chunk_build(size_t nelem, size_t nnode, const size_t* inpoel, const int* esuel,
                    const int32_t* part, int nparts, int rank, qdg_chunk** out);
/* The same with TWO ghost layers (depth = 2; depth = 1 is the call above): behind the face-neighbour ghosts
 * (layer 1) come the tets of other ranks that share a face with a layer-1 ghost (layer 2).  A rank then holds every
 * input of the limiter of its layer-1 ghosts (src/PDE/Limiter.cpp:29-316 read a tet's face neighbours) and limits
 * them itself, so the exchange of the LIMITED solution (DG::lim -> comlim, src/Inciter/DG.cpp:1262-1282) is not
 * needed: 3 exchanges per time step instead of 6 (qdg_halo_set_depth).  The plan then has one ENTRY per
 * (neighbour rank, layer): nnbr counts entries, nbr_rank[] lists the layer-1 entries (ranks ascending) followed
 * by the layer-2 entries (ranks ascending; a rank can appear in both, or in layer 2 only), each with its send
 * list and receive range as before; ghost rows = layer 1 (nghost1 rows, grouped by owner), then layer 2. */
int qdg_chunk_build_depth(size_t nelem, size_t nnode, const size_t* inpoel, const int* esuel,
                          const int32_t* part, int nparts, int rank, int depth, qdg_chunk** out);
int qdg_chunk_sizes(const qdg_chunk* c, size_t* nielem, size_t* nunk, size_t* nnode, size_t* nnbr,
                    size_t* nsend);
/* depth of the chunk, its number of layer-1 ghosts, the layer (1 / 2) of every plan entry (any pointer may be NULL) */
int qdg_chunk_layers(const qdg_chunk* c, int32_t* depth, size_t* nghost1, int32_t* nbr_layer);
/* The ghost layers and the halo plan ALONE, from the face adjacency esuel[4*nelem] of the tets around a rank's
 * own (complete up to `depth` faces away from them), their owner ranks and global ids (NULL: the index): for a
 * caller that assembles its chunk itself.  Tets are addressed by their index: ghost[nghost] = layer 1 (nghost1
 * tets, by owner and global id), then layer 2; entries as for qdg_chunk_build_depth (recv_off / send_off
 * [nentry + 1], send_elem[nsend] = owned tets). */
typedef struct qdg_ghost_plan qdg_ghost_plan;
int qdg_ghost_plan_build(size_t nelem, const int* esuel, const int32_t* owner, const size_t* gid, int rank,
                         int depth, qdg_ghost_plan** out);
int qdg_ghost_plan_sizes(const qdg_ghost_plan* p, size_t* nghost, size_t* nghost1, size_t* nentry, size_t* nsend);
int qdg_ghost_plan_get(const qdg_ghost_plan* p, size_t* ghost, int32_t* entry_rank, int32_t* entry_layer,
                       size_t* recv_off, size_t* send_off, size_t* send_elem);
int qdg_ghost_plan_destroy(qdg_ghost_plan* p);
/* copy-out (any pointer may be NULL): inpoel[4*nunk] local node ids, elem_gid[nunk],
 * node_gid[nnode], nbr_rank[nnbr], send_off[nnbr+1], send_elem[nsend] (local owned ids),
 * recv_off[nnbr+1] (ghost rows nielem + recv_off[i] ...) */
int qdg_chunk_get(const qdg_chunk* c, size_t* inpoel, size_t* elem_gid, size_t* node_gid,
                  int32_t* nbr_rank, size_t* send_off, size_t* send_elem, size_t* recv_off);
int qdg_chunk_destroy(qdg_chunk* c);

/* -- mesh refinement during time stepping (BASELINE config 5) ----------------------------
 * qdg_refine_uniform: uniform 1:8 refinement, the one refinement the reference's DG scheme
 * performs at t > 0 (Refiner::dtref with amr::dtref_uniform, src/Inciter/Refiner.cpp:403-408);
 * children as AMR::refinement_t::refine_one_to_eight builds them
 * (src/Inciter/AMR/refinement.hpp:425-536).  Result: 8*nelem tets (child 8*e+k of parent e),
 * the old nodes followed by the edge midpoints, parent[child], 4*ntri side-set triangles
 * (child 4*t+k of triangle t, same side set).  qdg_refined_get copies out (any pointer may be
 * NULL): inpoel[32*nelem], parent[8*nelem], x/y/z[nnode_new], tri[12*ntri].
 * qdg_state_transfer: the solution on the new mesh as DG::resizePostAMR sets it
 * (src/Inciter/DG.cpp:1597-1605): every child takes its parent's row (all DOFs), device to
 * device between two mesh handles of one context; parent_of_child in the caller's numbering
 * of both meshes (nunk(to) entries). */
typedef struct qdg_refined qdg_refined;
int qdg_refine_uniform(size_t nelem, size_t nnode, const size_t* inpoel, const double* x,
                       const double* y, const double* z, size_t ntri, const size_t* tri,
                       qdg_refined** out);
int qdg_refined_get(const qdg_refined* r, size_t* nnode, size_t* inpoel, size_t* parent,
                    double* x, double* y, double* z, size_t* tri);
int qdg_refined_destroy(qdg_refined* r);
/* Uniform 8:1 derefinement, the inverse of the above, for a mesh that is a uniform refinement in this library's order
 * (children 8 e + k, old nodes before the midpoints, child triangles 4 t + k): what the reference's Refiner does for
 * `uniform_derefine` at t0 (src/Inciter/Refiner.cpp:395-408, AMR/refinement.hpp:726-800; its t0ref goldens of uniform ->
 * uniform_derefine -> uniform hold the original mesh again after that step).  Result through qdg_refined_get: nelem/8
 * tets, the old nodes, ntri/4 triangles; parent[p] = 8 p (the first child of coarse tet p).  Any other input is
 * refused.  The reference's DG scheme never removes tets at t > 0 (src/Inciter/DG.cpp:1597-1605 handles added tets
 * only), so there is no reference for the solution transfer: qdg_mesh_derefine_uniform offers two. */
int qdg_derefine_uniform(size_t nelem, size_t nnode, const size_t* inpoel, const double* x, const double* y,
                         const double* z, size_t ntri, const size_t* tri, qdg_refined** out);
/* The same refinement computed on the context's GPU (one radix sort of the edge keys and two
 * scans) and copied back: identical arrays, for meshes whose host-side refinement time matters. */
int qdg_refine_uniform_device(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                              const double* x, const double* y, const double* z, size_t ntri,
                              const size_t* tri, qdg_refined** out);

/* Uniform refinement of ONE RANK's chunk of a decomposition, by the rank alone (the re-mesh step of
 * DG::resizePostAMR on a chare, src/Inciter/DG.cpp:1536-1612): in = the chunk as qdg_chunk_build
 * made it (owned tets [0, nielem), ghosts behind them grouped by owner in the order of nbr_rank
 * [ascending], gid = global tet ids, side-set triangles in local node ids, recv_counts per
 * neighbour).  Out: the children of the owned tets (8 * nielem, in order), the new ghost layer
 * (children of old ghosts that share a face with a new owned tet, grouped by owner, ordered by
 * global child id 8 * gid(parent) + k), nodes renumbered, side-set triangles of the refined
 * chunk, the new halo plan (send_off[nnbr + 1], send_list of owned local ids per neighbour ordered
 * by global child id; recv_counts[nnbr]) and parent[] = old local id of every kept child's
 * parent (for qdg_state_transfer).  No communication: both ranks of a pair derive the same
 * sets.  Follow with qdg_mesh_from_chunk + qdg_halo_setup + qdg_state_transfer. */
typedef struct qdg_chunk_refined qdg_chunk_refined;
int qdg_refine_chunk(size_t nielem, size_t nunk, size_t nnode, const size_t* inpoel, const double* x,
                     const double* y, const double* z, const size_t* gid, size_t ntri, const size_t* tri,
                     const int32_t* tri_set, size_t nnbr, const int32_t* nbr_rank,
                     const size_t* recv_counts, qdg_chunk_refined** out);
/* The same for a chunk with `depth` ghost layers (depth = 1: the call above; depth = 2: qdg_chunk_build_depth's
 * shape, nentry = the plan's (rank, layer) entries in order, recv_counts per entry).  The refined chunk's layers and
 * plan follow the rule of qdg_chunk_build_depth on the children (owner of a child = its parent's, global id
 * 8 * gid(parent) + k); its entries can differ from the old ones: qdg_chunk_refined_plan returns them (nentry,
 * nghost1, nbr_rank[nentry], nbr_layer[nentry]; NULL pointers: sizes only), qdg_chunk_refined_get's send_off /
 * recv_counts are sized by that nentry. */
int qdg_refine_chunk_depth(size_t nielem, size_t nunk, size_t nnode, const size_t* inpoel, const double* x,
                           const double* y, const double* z, const size_t* gid, size_t ntri, const size_t* tri,
                           const int32_t* tri_set, size_t nentry, const int32_t* entry_rank,
                           const size_t* recv_counts, int depth, qdg_chunk_refined** out);
int qdg_chunk_refined_plan(const qdg_chunk_refined* c, size_t* nentry, size_t* nghost1, int32_t* nbr_rank,
                           int32_t* nbr_layer);
int qdg_chunk_refined_sizes(const qdg_chunk_refined* c, size_t* nielem, size_t* nunk, size_t* nnode,
                            size_t* ntri, size_t* nsend);
int qdg_chunk_refined_get(const qdg_chunk_refined* c, size_t* inpoel, size_t* gid, size_t* parent,
                          double* x, double* y, double* z, size_t* tri, int32_t* tri_set,
                          size_t* send_off, size_t* send_list, size_t* recv_counts);
int qdg_chunk_refined_destroy(qdg_chunk_refined* c);
int qdg_state_transfer(qdg_mesh* from, qdg_mesh* to, const size_t* parent_of_child);
/* The whole re-mesh of a resident chunk WITHOUT ghosts in one call, with nothing but the host's copy of the
 * refined mesh crossing PCIe (DG::resizePostAMR, src/Inciter/DG.cpp:1536-1612; BASELINE config 5): uniform
 * 1:8 refinement from the connectivity `mesh` kept on the device (build it with context option
 * "keep_connectivity" = 1; a mesh made by this call always keeps its own), esuel of the children derived from
 * the parents' by the 1:8 template (src/Inciter/AMR/refinement.hpp:425-536) instead of a sort over all child
 * faces, boundary faces from the children of the parent's boundary faces, the same device layout as every
 * other build (the new handle is array for array the one qdg_mesh_from_connectivity makes from the refined
 * connectivity), and the state handed over child <- parent.  `mesh` stays valid (destroy it when done).
 * host_copy (may be NULL): the refined mesh for the host's book-keeping -- children 8 e + k of tet e, old
 * nodes followed by edge midpoints as qdg_refine_uniform numbers them, triangles = the children 4 b + k of
 * the parent's boundary faces b in the library's order (side set: qdg_refined_tri_sets) -- filled by a second
 * host thread on a second stream while this call builds the mesh; qdg_refined_sizes / _get / _tri_sets wait
 * for it. */
int qdg_mesh_refine_uniform(qdg_mesh* mesh, qdg_mesh** refined_mesh, qdg_refined** host_copy);
/* ... and of ONE RANK's chunk WITH its ghost layer (config 5 on a decomposition; the device form of
 * qdg_refine_chunk): the handle was built by qdg_mesh_from_chunk_gid (the tets' global ids order the new ghosts)
 * under "keep_connectivity" = 1 and has had its qdg_halo_setup.  Children of the owned tets, the new ghost layer
 * (children of old ghosts that share a face with an owned child, grouped by owner, by global child id
 * 8 * gid(parent) + k), the new halo plan and the renumbered nodes are derived on the device exactly as
 * qdg_refine_chunk derives them on the host -- without communication: both ranks of a pair get the same sets --
 * then the chunk build, qdg_halo_setup of the new handle and the state of the owned tets (child <- parent).
 * host_copy (may be NULL): gid[nunk], parent[nunk] (old local id), send lists, receive counts -- and with
 * copy_mesh != 0 connectivity, coordinates and side-set triangles too -- through qdg_chunk_refined_sizes / _get.
 * Limits: fewer than 65 536 neighbour ranks and global child ids below 2^48 (the device sorts owner << 48 | id). */
/* (A handle with two ghost layers -- qdg_halo_set_depth -- is re-meshed with two: the children within two faces of the
 * old owned | ghost interface are cut out on the device, the rule of qdg_chunk_build_depth derives the new layers and
 * plan from them, qdg_halo_setup + qdg_halo_set_depth of the new handle included; its plan entries:
 * qdg_chunk_refined_plan.) */
int qdg_mesh_refine_chunk(qdg_mesh* mesh, qdg_mesh** refined_mesh, qdg_chunk_refined** host_copy, int copy_mesh);
/* The inverse for a resident chunk without ghosts whose kept connectivity is a uniform refinement in this library's
 * order (qdg_derefine_uniform): 8 children -> their parent, on the device; the coarse boundary faces follow from the
 * refined ones (same side sets), then the general device build.  There is no reference for the solution of a removed
 * tet (src/Inciter/DG.cpp:1597-1605 handles added tets only): policy QDG_DEREF_FIRST_CHILD gives a parent the row of its
 * first child -- the exact inverse of the reference's row copy child <- parent, so refinement followed by derefinement
 * returns every DOF --, QDG_DEREF_MEAN the volume-weighted mean of the children's means with zero higher-order DOFs
 * (conservative).  The new handle keeps its connectivity (it can be refined again). */
enum { QDG_DEREF_FIRST_CHILD = 0, QDG_DEREF_MEAN = 1 };
int qdg_mesh_derefine_uniform(qdg_mesh* mesh, int policy, qdg_mesh** coarse_mesh);
int qdg_refined_sizes(const qdg_refined* r, size_t* nelem, size_t* nnode, size_t* ntri);
int qdg_refined_tri_sets(const qdg_refined* r, int32_t* tri_set);
/* The child mesh need not come from qdg_refine_*: ANY conforming tetrahedron mesh with a parent
 * per tet is accepted -- what Refiner hands DG::resizePostAMR (src/Inciter/DG.cpp:1537-1612), e.g. a
 * region refined 1:8 and closed with the 1:2 / 1:4 templates of src/Inciter/AMR/refinement.hpp:78-424;
 * build `to` with qdg_mesh_from_connectivity / qdg_mesh_from_chunk.  An entry QDG_NO_ROW leaves
 * that row of `to` as it is.
 *
 * State migration after a re-partition (the load balancing behind DG::resizePostAMR,
 * src/Inciter/DG.cpp:1658-1664, Partitioner.cpp:137-170): rows travel by GLOBAL tet id.
 *   qdg_state_migrate   both chunks in one context (one GPU): every owned row of `to` whose
 *                       global id is owned by `from` is copied device to device; call it for
 *                       every (from, to) pair; *nmoved (may be NULL) = rows copied
 *   qdg_state_rows_get / _put   the two halves for chunks of DIFFERENT processes: n rows (caller's
 *                       numbering) <-> a packed device buffer of n * nprop doubles that the caller
 *                       moves (ncclSend / ncclRecv, MPI, hipMemcpyPeer) */
#define QDG_NO_ROW ((size_t)-1)
int qdg_state_migrate(qdg_mesh* from, const size_t* from_gid, qdg_mesh* to, const size_t* to_gid,
                      size_t* nmoved);
int qdg_state_rows_get(qdg_mesh* mesh, size_t n, const size_t* rows, void* packed_dev);
int qdg_state_rows_put(qdg_mesh* mesh, size_t n, const size_t* rows, const void* packed_dev);

/* -- element-field output in ExodusII layout -----------------------------------
 * What DG::writeFields hands tk::ExodusIIMeshWriter (src/Inciter/DG.cpp:1165-1215,
 * src/IO/ExodusIIMeshWriter.cpp: writeMesh, writeElemVarNames, writeTimeStamp, writeElemScalar)
 * written as a netCDF classic file with 64-bit offsets (CDF-2) in ExodusII conventions -- one
 * TETRA block, side sets as (element, ExodusII side 1..4) pairs, element variables -- so that the
 * reference's exodiff configuration (exodiff_dg.cfg) can compare it with a golden file.
 * vals[(t*nvar + v)*nelem + e]; all time steps are written by the one call. */
int qdg_exo_write(const char* path, const char* title, size_t nnode, const double* x, const double* y,
                  const double* z, size_t nelem, const size_t* inpoel, size_t nss, const int32_t* ss_id,
                  const size_t* ss_off, const size_t* ss_elem, const int32_t* ss_side, size_t nvar,
                  const char* const* var_names, size_t ntime, const double* times, const double* vals);

/* -- mesh-derived data generated on the device (first step) -----
 * The same arrays as qdg_gen_esuel / nipfac / inpofa / belem / esuf / geoface /
 * geoelem above (src/Inciter/FaceData.cpp:19-41, src/Mesh/DerivedData.cpp:937-1491),
 * same content and order, computed by sort/scan kernels on the context's GPU and
 * copied back.  Capacities: esuel 4*nelem, inpofa 3*(nbfac+2*nelem),
 * esuf 2*(nbfac+2*nelem), belem nbfac, geoFace 7*(nbfac+2*nelem), geoElem 4*nelem;
 * *nipfac returns the number of faces (boundary + interior). */
int qdg_dev_facedata(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                     const double* x, const double* y, const double* z, size_t nbfac,
                     const size_t* triinpoel, int* esuel, size_t* nipfac, size_t* inpofa,
                     int* esuf, size_t* belem, double* geoFace, double* geoElem);

/* One call from connectivity to a mesh handle, for a chunk WITHOUT ghosts (serial run,
 * or the re-build after mesh refinement, DG::resizePostAMR src/Inciter/DG.cpp:1536-1612):
 * boundary faces regenerated from the side-set triangles (the order of qdg_bnd_faces), FaceData
 * and geometry (the arrays of qdg_dev_facedata), device order, numbering and face tasks (what
 * qdg_mesh_upload derives on the host) -- all on the device; only connectivity, coordinates and
 * the side-set triangles cross PCIe.  tri_set[i] is the side set id of triangle i; a triangle listed in
 * several side sets belongs to the one with the highest id, as in the reference's loader (the last writer of
 * `faceside`, src/Inciter/Partitioner.cpp:358-364), and is integrated once (src/PDE/Integrate/Boundary.cpp:84-86).
 * (context option "host_layout" = 1: FaceData on the device, layout by qdg_mesh_upload, for
 * equivalence tests.) */
int qdg_mesh_from_connectivity(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                               const double* x, const double* y, const double* z, size_t ntri,
                               const size_t* tri, const int32_t* tri_set, qdg_mesh** out);

/* The same for one rank's chunk WITH its ghost layer (what the DG chare holds after its ghost
 * set-up, src/Inciter/DG.cpp:468-712; after a re-mesh on a decomposition, DG.cpp:1536-1612):
 * tets [0, nielem) are owned, [nielem, nelem) are the face-neighbour ghosts in the order of the
 * halo plan (qdg_chunk_build), nodes and coordinates cover both.  Everything is derived on the
 * device: boundary faces of the OWNED tets from the side-set triangles, interior and
 * chare-boundary faces (a face between two ghosts is none of this chunk's), geometry, the
 * device order (tets with a ghost neighbour last: the halo's send rows are the trailing rows),
 * numbering, face tasks.  Follow with qdg_halo_setup as after qdg_mesh_upload.
 * nielem == nelem: identical to qdg_mesh_from_connectivity. */
int qdg_mesh_from_chunk(qdg_ctx* ctx, size_t nielem, size_t nelem, size_t nnode, const size_t* inpoel,
                        const double* x, const double* y, const double* z, size_t ntri,
                        const size_t* tri, const int32_t* tri_set, qdg_mesh** out);
/* ... with the tets' global ids (elem_gid[nelem], e.g. qdg_chunk_get's elem_gid; NULL = the call above):
 * faces oriented by global id as described at qdg_mesh_upload_gid, the face's nodes taken in the new
 * left tet's local face order and the geometry computed from them, as the serial run stores the face. */
int qdg_mesh_from_chunk_gid(qdg_ctx* ctx, size_t nielem, size_t nelem, size_t nnode, const size_t* inpoel,
                            const double* x, const double* y, const double* z, size_t ntri,
                            const size_t* tri, const int32_t* tri_set, const size_t* elem_gid,
                            qdg_mesh** out);

#ifdef __cplusplus
}
#endif
#endif /* QDG_H */
