// qdg_api.cpp -- host layer of libqdg: the C ABI of include/qdg.h on top of the
// gfx950 kernels (qdg_kernels.hip).  Mesh upload builds the device layout
// described in qdg_device.hpp; every operator is either stateless on host
// `tk::Fields`-layout arrays (the DGPDE-shaped calls) or runs on the
// device-resident state.  There is no CPU fallback in this library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen'ed (see rccl_api)
#include <dlfcn.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <atomic>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/qdg.h"
#include "qdg_device.hpp"
#include "qdg_host.hpp"
#include "qdg_handles.hpp"
#include "qdg_kernels.hpp"
#include "qdg_tables.hpp"

namespace qdg {

static thread_local std::string g_err;
void set_error(const std::string& m) { g_err = m; }
int fail(const std::string& m) { g_err = m; return 1; }

#define HIPCHK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess)                                                         \
      return ::qdg::fail(std::string(#call) + ": " + hipGetErrorString(e_));      \
  } while (0)

// run fn(begin, end) over [0, n) on the host's cores (mesh set-up loops)
template <class F> static void parallel_for(size_t n, F&& fn, size_t serial_below = 65536)
{
  unsigned nt = std::thread::hardware_concurrency();
  nt = std::max(1u, std::min(nt, 16u));
  if (n < serial_below || nt == 1) { fn((size_t)0, n); return; }
  std::vector<std::thread> th;
  const size_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t b = std::min(n, t * chunk), e = std::min(n, b + chunk);
    if (b < e) th.emplace_back([&fn, b, e] { fn(b, e); });
  }
  for (auto& t : th) t.join();
}

}  // namespace qdg

using namespace qdg;

namespace qdg {
// for the other translation units of the library (qdg_devmesh.hip)
int ctx_device(const qdg_ctx* ctx) { return ctx->device; }
hipStream_t ctx_stream(const qdg_ctx* ctx) { return ctx->stream; }
}  // namespace qdg

// ---------------------------------------------------------------- misc

extern "C" const char* qdg_last_error(void) { return g_err.c_str(); }
extern "C" const char* qdg_version(void) { return "quinoa_amd/qdg 0.1 (gfx950)"; }

static int check_cfg(const qdg_config* c)
{
  if (!c) return fail("qdg_ctx_create: null config");
  if (c->struct_size != (int32_t)sizeof(qdg_config))
    return fail("qdg_ctx_create: qdg_config.struct_size mismatch (ABI)");
  if (!(c->ndof == 1 || c->ndof == 4 || c->ndof == 10))
    return fail("qdg_ctx_create: ndof must be one of 1,4,10");
  if (c->rdof != c->ndof)
    return fail("qdg_ctx_create: rdof != ndof (P0P1 reconstruction) is not supported");
  if (c->limiter < QDG_LIMITER_NONE || c->limiter > QDG_LIMITER_SUPERBEEP1)
    return fail("qdg_ctx_create: unknown limiter");
  if (c->nbc < 0 || (c->nbc > 0 && (!c->bc_sideset || !c->bc_type)))
    return fail("qdg_ctx_create: bad BC table");
  if (c->pde == QDG_PDE_TRANSPORT) {
    if (c->flux != QDG_FLUX_UPWIND) return fail("qdg_ctx_create: transport needs the upwind flux");
    if (!(c->problem == QDG_PROBLEM_SLOT_CYL || c->problem == QDG_PROBLEM_CYL_ADVECT ||
          c->problem == QDG_PROBLEM_GAUSS_HUMP || c->problem == QDG_PROBLEM_SHEAR_DIFF))
      return fail("qdg_ctx_create: unknown transport problem");
    if (c->ncomp < 0 || c->ncomp > QDG_MAX_SCALARS)
      return fail("qdg_ctx_create: transport ncomp must be 1.." + std::to_string(QDG_MAX_SCALARS));
    if (c->problem == QDG_PROBLEM_SHEAR_DIFF) {
      // TransportProblemShearDiff::errchk (ShearDiff.cpp:92-113): u0, lambda, diffusivity per scalar
      if (!c->tr_u0 || !c->tr_lambda || !c->tr_diffusivity)
        return fail("qdg_ctx_create: shear_diff needs u0[ncomp], lambda[2*ncomp], diffusivity[3*ncomp]");
      for (int i = 0; i < 3 * (c->ncomp ? c->ncomp : 1); ++i)
        if (!(c->tr_diffusivity[i] > 0.0)) return fail("qdg_ctx_create: shear_diff diffusivity must be > 0");
    }
    if (c->pref && c->ncomp > 1)
      return fail("qdg_ctx_create: p-adaptive DG is supported for one transported scalar");
    if (!(c->dt > 0.0))
      return fail("qdg_ctx_create: transport needs a constant dt (dg::Transport::dt gives no CFL estimate)");
    if (c->pref) {
      if (c->ndof != 4) return fail("qdg_ctx_create: p-adaptive DG needs ndof = rdof = 4");
      if (c->limiter == QDG_LIMITER_WENOP1) return fail("qdg_ctx_create: p-adaptive DG with WENO is not supported");
      if (!(c->tolref >= 0.0)) return fail("qdg_ctx_create: bad tolref");
    }
    for (int i = 0; i < c->nbc; ++i) {
      const int b = c->bc_type[i];
      if (!(b == QDG_BC_DIRICHLET || b == QDG_BC_EXTRAPOLATE || b == QDG_BC_INLET || b == QDG_BC_OUTLET))
        return fail("qdg_ctx_create: unknown transport BC type");
    }
    return 0;
  }
  if (c->pde != QDG_PDE_COMPFLOW) return fail("qdg_ctx_create: unknown pde");
  if (!(c->ncomp == 0 || c->ncomp == NCOMP)) return fail("qdg_ctx_create: CompFlow has 5 components");
  if (c->pref) {
    if (c->ndof != 4) return fail("qdg_ctx_create: p-adaptive DG needs ndof = rdof = 4");
    if (c->limiter == QDG_LIMITER_WENOP1) return fail("qdg_ctx_create: p-adaptive DG with WENO is not supported");
    if (!(c->tolref >= 0.0)) return fail("qdg_ctx_create: bad tolref");
  }
  if (c->flux != QDG_FLUX_HLLC && c->flux != QDG_FLUX_LAXFRIEDRICHS)
    return fail("qdg_ctx_create: unknown flux");
  if (!((c->problem >= QDG_PROBLEM_USER_DEFINED && c->problem <= QDG_PROBLEM_TAYLOR_GREEN) ||
        c->problem == QDG_PROBLEM_ROTATED_SOD_SHOCKTUBE || c->problem == QDG_PROBLEM_NL_ENERGY_GROWTH ||
        c->problem == QDG_PROBLEM_RAYLEIGH_TAYLOR))   /* 5, 8, 9: transport */
    return fail("qdg_ctx_create: unknown problem");
  if (!(c->gamma > 1.0)) return fail("qdg_ctx_create: gamma must be > 1");
  for (int i = 0; i < c->nbc; ++i)
    if (c->bc_type[i] < QDG_BC_DIRICHLET || c->bc_type[i] > QDG_BC_EXTRAPOLATE)
      return fail("qdg_ctx_create: unknown BC type");
  return 0;
}

extern "C" int qdg_ctx_create(const qdg_config* cfg, qdg_ctx** out)
{
  QDG_TRY
  if (!out) return fail("qdg_ctx_create: null out");
  *out = nullptr;
  if (int rc = check_cfg(cfg)) return rc;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail("qdg_ctx_create: no HIP device available (this library has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail("qdg_ctx_create: bad device ordinal");
  HIPCHK(hipSetDevice(cfg->device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(std::string("qdg_ctx_create: device is ") + prop.gcnArchName +
                ", this library is built for gfx950 (MI355X) only");
  std::unique_ptr<qdg_ctx> c(new qdg_ctx);
  c->cfg = *cfg;
  c->bc_sideset.assign(cfg->bc_sideset, cfg->bc_sideset + cfg->nbc);
  c->bc_type.assign(cfg->bc_type, cfg->bc_type + cfg->nbc);
  c->cfg.bc_sideset = c->bc_sideset.data();
  c->cfg.bc_type = c->bc_type.data();
  c->device = cfg->device;
  c->ph.gamma = cfg->gamma; c->ph.pstiff = cfg->pstiff; c->ph.cweight = cfg->cweight;
  c->ph.cv = cfg->cv;
  c->ph.alpha = cfg->alpha; c->ph.beta = cfg->beta; c->ph.p0 = cfg->p0;
  c->ph.betax = cfg->betax; c->ph.betay = cfg->betay; c->ph.betaz = cfg->betaz;
  c->ph.r0 = cfg->r0; c->ph.ce = cfg->ce; c->ph.kappa = cfg->kappa;
  c->ph.flux = cfg->flux; c->ph.problem = cfg->problem; c->ph.limiter = cfg->limiter;
  for (double& v : c->ph.sd_u0) v = 0.0;
  for (double& v : c->ph.sd_lambda) v = 0.0;
  for (double& v : c->ph.sd_diff) v = 1.0;
  if (cfg->pde == QDG_PDE_TRANSPORT && cfg->problem == QDG_PROBLEM_SHEAR_DIFF) {
    const int nc = cfg->ncomp ? cfg->ncomp : 1;
    for (int i = 0; i < nc; ++i) c->ph.sd_u0[i] = cfg->tr_u0[i];
    for (int i = 0; i < 2 * nc; ++i) c->ph.sd_lambda[i] = cfg->tr_lambda[i];
    for (int i = 0; i < 3 * nc; ++i) c->ph.sd_diff[i] = cfg->tr_diffusivity[i];
  }
  c->cfg.tr_u0 = c->cfg.tr_lambda = c->cfg.tr_diffusivity = nullptr;   // copied into ph: no pointer into the caller kept
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  c->own_stream = true;
  // constant tables (one copy per device/module)
  {
    auto t1 = std::make_unique<Tables<1>>();
    auto t4 = std::make_unique<Tables<4>>();
    auto t10 = std::make_unique<Tables<10>>();
    fill_tables(*t1); fill_tables(*t4); fill_tables(*t10);
    QuadTet qi[3], qd[3];
    const int nd[3] = { 1, 4, 10 };
    for (int i = 0; i < 3; ++i) { fill_quadtet(qi[i], nginit(nd[i])); fill_quadtet(qd[i], ngdiag(nd[i])); }
    HIPCHK(upload_tables(*t1, *t4, *t10, qi, qd));
  }
  qdg::DevicePool::get().ctx_opened();
  *out = c.release();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_ctx_destroy(qdg_ctx* ctx)
{
  QDG_TRY
  if (!ctx) return 0;
  const bool keep = ctx->opt.keep_pool != 0;
  delete ctx;      // ~qdg_ctx drains and destroys the stream it owns
  // the process's last context returns the device buffer cache (an embedding application shares the GPU
  // with other allocators that never see this cache); option keep_pool = 1 keeps it
  if (qdg::DevicePool::get().ctx_closed() == 0 && !keep) (void)qdg::DevicePool::get().trim();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_device_pool_trim(size_t* released_bytes)
{
  QDG_TRY
  const size_t n = qdg::DevicePool::get().trim();
  if (released_bytes) *released_bytes = n;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_device_pool_reserve(qdg_ctx* ctx, size_t bytes)
{
  QDG_TRY
  if (!ctx) return fail("qdg_device_pool_reserve: null ctx");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(qdg::DevicePool::get().reserve(bytes));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_device_memory(qdg_ctx* ctx, size_t* free_bytes, size_t* total_bytes, size_t* reserved_bytes)
{
  QDG_TRY
  if (!ctx) return fail("qdg_device_memory: null ctx");
  HIPCHK(hipSetDevice(ctx->device));
  size_t fr = 0, tot = 0;
  HIPCHK(hipMemGetInfo(&fr, &tot));
  if (free_bytes) *free_bytes = fr;
  if (total_bytes) *total_bytes = tot;
  if (reserved_bytes) *reserved_bytes = qdg::DevicePool::get().reserved_bytes();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_device_alloc(qdg_ctx* ctx, size_t bytes, void** out)
{
  QDG_TRY
  if (!ctx || !out) return fail("qdg_device_alloc: null argument");
  *out = nullptr;
  if (bytes == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  // (no stream tag: the caller may use the buffer on any stream, so a recycled block waits for the device)
  HIPCHK(qdg::dev_alloc(out, bytes));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_device_free(qdg_ctx* ctx, void* ptr)
{
  QDG_TRY
  if (!ctx) return fail("qdg_device_free: null ctx");
  if (!ptr) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  // the caller may have used the buffer on any stream: freed without a stream tag, the block waits for the
  // device before it is handed out again (as hipFree would have)
  qdg::dev_free(ptr);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_ctx_set_stream(qdg_ctx* ctx, void* s)
{
  QDG_TRY
  if (!ctx) return fail("qdg_ctx_set_stream: null ctx");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->own_stream && ctx->stream) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const hipStream_t old = ctx->stream;
    ctx->stream = nullptr; ctx->own_stream = false;
    HIPCHK(hipStreamDestroy(old));
  }
  ctx->stream = (hipStream_t)s;
  ctx->own_stream = false;
  return 0;
  QDG_CATCH
}

// components of a row: 5 (CompFlow) or the transported scalars of the dg::Transport system
static inline int ctx_ncomp(const qdg_ctx* ctx)
{
  return ctx->cfg.pde == QDG_PDE_TRANSPORT ? (ctx->cfg.ncomp > 0 ? ctx->cfg.ncomp : 1) : NCOMP;
}

// tuning / A-B switches by name (no behaviour is read from the process environment)
static int* option_slot(qdg_ctx* ctx, const char* name)
{
  static const struct { const char* n; int qdg::Options::*p; } tab[] = {
    { "p1_rhs", &qdg::Options::p1_rhs },
    { "fused_update", &qdg::Options::fused_update },
    { "renumber", &qdg::Options::renumber },       { "host_layout", &qdg::Options::host_layout },
    { "orient_by_gid", &qdg::Options::orient_by_gid }, { "keep_pool", &qdg::Options::keep_pool },
    { "keep_connectivity", &qdg::Options::keep_connectivity },
    { "graph_step", &qdg::Options::graph_step }, { "halo_depth", &qdg::Options::halo_depth },
    { "limiter_write_all", &qdg::Options::limiter_write_all },
  };
  for (const auto& t : tab)
    if (std::strcmp(name, t.n) == 0) return &(ctx->opt.*(t.p));
  return nullptr;
}

extern "C" int qdg_ctx_set_option(qdg_ctx* ctx, const char* name, int value)
{
  QDG_TRY
  if (!ctx || !name) return fail("qdg_ctx_set_option: null argument");
  int* p = option_slot(ctx, name);
  if (!p) return fail(std::string("qdg_ctx_set_option: unknown option '") + name + "'");
  if (p == &ctx->opt.p1_rhs && value != 0 && value != 1)
    return fail("qdg_ctx_set_option: p1_rhs is 0 (tile / face-task kernel) or 1 (element-centric, reproducible)");
  *p = value;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_ctx_get_option(qdg_ctx* ctx, const char* name, int* value)
{
  QDG_TRY
  if (!ctx || !name || !value) return fail("qdg_ctx_get_option: null argument");
  const int* p = option_slot(ctx, name);
  if (!p) return fail(std::string("qdg_ctx_get_option: unknown option '") + name + "'");
  *value = *p;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_ctx_synchronize(qdg_ctx* ctx)
{
  QDG_TRY
  if (!ctx) return fail("qdg_ctx_synchronize: null ctx");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_solution(qdg_ctx* ctx, size_t n, const double* x, const double* y, const double* z,
                            double t, double* out)
{
  QDG_TRY
  if (!ctx || (n && (!x || !y || !z || !out))) return fail("qdg_solution: null argument");
  if (n > (size_t)INT32_MAX / 8) return fail("qdg_solution: too many points for one call");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int ncomp = ctx_ncomp(ctx);
  DevBuf<double> d;
  HIPCHK(d.alloc((3 + (size_t)ncomp) * n));
  HIPCHK(hipMemcpyAsync(d.p, x, n * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d.p + n, y, n * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(d.p + 2 * n, z, n * 8, hipMemcpyHostToDevice, s));
  launch_solution(ctx->cfg.pde, ncomp, ctx->ph, (int)n, d.p, d.p + n, d.p + 2 * n, t, d.p + 3 * n, s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, d.p + 3 * n, (size_t)ncomp * n * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

namespace qdg {
// fields and work buffers of a mesh handle whose sizes (nie, ne, stride, nprop) are set
int mesh_alloc_state(qdg_mesh* m, int ntile)
{
  hipStream_t s = m->ctx->stream;
  const size_t fsz = (size_t)m->nprop * m->stride;
  HIPCHK(m->U.alloc(fsz)); HIPCHK(m->Un.alloc(fsz)); HIPCHK(m->R.alloc(fsz)); HIPCHK(m->W.alloc(fsz));
  // (m->aos, the staging buffer of rows in the caller's numbering, is allocated by its first user:
  // a resident loop that never moves the state to the host does not pay for it)
  HIPCHK(hipMemsetAsync(m->U.p, 0, fsz * sizeof(double), s));
  HIPCHK(hipMemsetAsync(m->Un.p, 0, fsz * sizeof(double), s));
  HIPCHK(hipMemsetAsync(m->R.p, 0, fsz * sizeof(double), s));
  HIPCHK(hipMemsetAsync(m->W.p, 0, fsz * sizeof(double), s));
  const size_t nblk = std::max((m->nie + 127) / 128, (size_t)ntile);   // k_rhs_p2s: 128 tets per workgroup
  HIPCHK(m->blockmin.alloc(nblk)); HIPCHK(m->dtraw.alloc(1)); HIPCHK(m->dtdev.alloc(1));
  // (dg::Transport: one set of block results per scalar)
  HIPCHK(m->diagpart.alloc(nblk * 15 * QDG_MAX_SCALARS)); HIPCHK(m->diagout.alloc(15));
  m->Ucur = m->U.p;
  m->dt_ptr = m->dtdev.p;
  return 0;
}
}  // namespace qdg

// ---------------------------------------------------------------- upload

static inline uint64_t spread21(uint64_t v)
{
  v &= 0x1fffff;
  v = (v | v << 32) & 0x1f00000000ffffULL;
  v = (v | v << 16) & 0x1f0000ff0000ffULL;
  v = (v | v << 8) & 0x100f00f00f00f00fULL;
  v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
  v = (v | v << 2) & 0x1249249249249249ULL;
  return v;
}

extern "C" int qdg_mesh_upload(qdg_ctx* ctx, size_t nielem, size_t nunk, size_t nnode,
                               const size_t* inpoel, const double* x, const double* y,
                               const double* z, size_t nbfac, size_t nfac, const int* esuf,
                               const int* esuel, const size_t* inpofa, const double* geoFace,
                               const double* geoElem, const qdg_bface* bface, qdg_mesh** out)
{
  return qdg_mesh_upload_gid(ctx, nielem, nunk, nnode, inpoel, x, y, z, nbfac, nfac, esuf, esuel, inpofa, geoFace,
                             geoElem, bface, nullptr, out);
}

extern "C" int qdg_mesh_upload_gid(qdg_ctx* ctx, size_t nielem, size_t nunk, size_t nnode,
                                   const size_t* inpoel, const double* x, const double* y,
                                   const double* z, size_t nbfac, size_t nfac, const int* esuf,
                                   const int* esuel, const size_t* inpofa, const double* geoFace,
                                   const double* geoElem, const qdg_bface* bface, const size_t* elem_gid,
                                   qdg_mesh** out)
{
  QDG_TRY
  if (!ctx || !out) return fail("qdg_mesh_upload: null ctx/out");
  *out = nullptr;
  if (!inpoel || !x || !y || !z || !esuf || !esuel || !inpofa || !geoFace || !geoElem)
    return fail("qdg_mesh_upload: null mesh array");
  if (nielem == 0 || nunk < nielem) return fail("qdg_mesh_upload: need 0 < nielem <= nunk");
  if (nunk > (size_t)(INT32_MAX - 64) / 4 || nnode > (size_t)INT32_MAX || nfac > (size_t)INT32_MAX)
    return fail("qdg_mesh_upload: chunk too large for 32-bit device indices");
  if (nbfac > nfac) return fail("qdg_mesh_upload: nbfac > nfac");
  HIPCHK(hipSetDevice(ctx->device));
  qdg::StreamScope scope(ctx->stream);
  const size_t nie = nielem, ne = nunk;

  // QDG_UPLOAD_STATS=1: wall time of the host sections below, on stderr
  const bool stats = std::getenv("QDG_UPLOAD_STATS") != nullptr;
  auto tprev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!stats) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "qdg upload: %-34s %7.1f ms\n", what,
                 std::chrono::duration<double, std::milli>(now - tprev).count());
    tprev = now;
  };

  // ---- validate connectivity before anything reaches a kernel -----------
  for (size_t i = 0; i < 4 * ne; ++i)
    if (inpoel[i] >= nnode) return fail("qdg_mesh_upload: inpoel entry out of range");
  for (size_t i = 0; i < 4 * nie; ++i)
    if (esuel[i] < -1 || (size_t)(esuel[i] + 1) > ne) return fail("qdg_mesh_upload: esuel entry out of range");
  for (size_t e = 0; e < ne; ++e)
    if (!(geoElem[4 * e] > 0.0)) return fail("qdg_mesh_upload: non-positive element volume");

  lap("validation");
  // ---- device order of interior tets: Morton curve of the centroids ------
  std::vector<int> d2h(ne), h2d(ne);
  std::iota(d2h.begin(), d2h.end(), 0);
  if (ctx->opt.renumber) {
    double lo[3] = { DBL_MAX, DBL_MAX, DBL_MAX }, hi[3] = { -DBL_MAX, -DBL_MAX, -DBL_MAX };
    for (size_t e = 0; e < nie; ++e)
      for (int d = 0; d < 3; ++d) {
        lo[d] = std::min(lo[d], geoElem[4 * e + 1 + d]);
        hi[d] = std::max(hi[d], geoElem[4 * e + 1 + d]);
      }
    const double ext = std::max({ hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300 });
    std::vector<std::pair<uint64_t, int>> key(nie);
    for (size_t e = 0; e < nie; ++e) {
      uint64_t k = 0;
      for (int d = 0; d < 3; ++d) {
        const double t = (geoElem[4 * e + 1 + d] - lo[d]) / ext;
        const uint64_t q = (uint64_t)std::min(2097151.0, std::max(0.0, t * 2097152.0));
        k |= spread21(q) << d;
      }
      key[e] = { k, (int)e };
    }
    std::sort(key.begin(), key.end());
    for (size_t d = 0; d < nie; ++d) d2h[d] = key[d].second;
  }
  // tets with a ghost neighbour go last (still in curve order): launches over the
  // leading rows never touch the halo (the send rows of the halo exchange are the trailing ones)
  size_t ninner = nie;
  if (ne > nie) {
    auto at_halo = [&](int e) {
      for (int lf = 0; lf < 4; ++lf) if (esuel[4 * (size_t)e + lf] >= (int)nie) return true;
      return false;
    };
    ninner = std::stable_partition(d2h.begin(), d2h.begin() + nie, [&](int e) { return !at_halo(e); }) - d2h.begin();
  }
  for (size_t d = 0; d < ne; ++d) h2d[d2h[d]] = (int)d;

  lap("curve order");
  // ---- nodes renumbered by first touch in device order -------------------
  std::vector<int> nnew(nnode, -1);
  int ncount = 0;
  for (size_t d = 0; d < ne; ++d)
    for (int i = 0; i < 4; ++i) {
      const size_t n = inpoel[4 * (size_t)d2h[d] + i];
      if (nnew[n] < 0) nnew[n] = ncount++;
    }
  std::vector<double> hx(ncount), hy(ncount), hz(ncount);
  for (size_t n = 0; n < nnode; ++n)
    if (nnew[n] >= 0) { hx[nnew[n]] = x[n]; hy[nnew[n]] = y[n]; hz[nnew[n]] = z[n]; }

  lap("node renumbering");
  // ---- (host element, local face) -> reference face id --------------------
  std::vector<int> rface(4 * nie, -1);
  for (size_t f = 0; f < nfac; ++f) {
    const int el = esuf[2 * f], er = esuf[2 * f + 1];
    if (el < 0 || (size_t)el >= nie) return fail("qdg_mesh_upload: esuf left element is not an interior tet");
    if (er < -1 || (er >= 0 && (size_t)er >= ne)) return fail("qdg_mesh_upload: esuf right element out of range");
    if (er == -1) {
      if (f >= nbfac) return fail("qdg_mesh_upload: boundary face beyond nbfac");
      int found = -1;
      for (int lf = 0; lf < 4 && found < 0; ++lf) {
        int cnt = 0;
        for (int j = 0; j < 3; ++j)
          for (int k = 0; k < 3; ++k)
            if (inpoel[4 * (size_t)el + LPOFA[lf][j]] == inpofa[3 * f + k]) ++cnt;
        if (cnt == 3) found = lf;
      }
      if (found < 0) return fail("qdg_mesh_upload: boundary face nodes do not match its element");
      if (esuel[4 * (size_t)el + found] != -1)
        return fail("qdg_mesh_upload: boundary face on an element face that has a neighbour");
      rface[4 * (size_t)el + found] = (int)f;
    } else {
      if (f < nbfac) return fail("qdg_mesh_upload: interior face inside the boundary-face range");
      int a = -1, b = -1;
      for (int lf = 0; lf < 4; ++lf) if (esuel[4 * (size_t)el + lf] == er) a = lf;
      if (a < 0) return fail("qdg_mesh_upload: esuf/esuel mismatch (left)");
      rface[4 * (size_t)el + a] = (int)f;
      if ((size_t)er < nie) {
        for (int lf = 0; lf < 4; ++lf) if (esuel[4 * (size_t)er + lf] == el) b = lf;
        if (b < 0) return fail("qdg_mesh_upload: esuf/esuel mismatch (right)");
        rface[4 * (size_t)er + b] = (int)f;
      }
    }
  }
  for (size_t i = 0; i < 4 * nie; ++i)
    if (rface[i] < 0) return fail("qdg_mesh_upload: element face without an entry in esuf "
                                  "(every boundary face must be listed in [0,nbfac))");

  // ---- orientation by global tet id (option orient_by_gid; see k_orient_gid in qdg_devmesh.hip) ----
  // flip[f]: the face's stored left tet (the caller's esuf[2f], an owned tet) has the HIGHER global id:
  // the device mesh takes the other tet as left and the negated normal, as the serial run of the whole
  // mesh stores the face (src/Mesh/DerivedData.cpp:1127-1139)
  std::vector<char> flip;
  if (elem_gid && ctx->opt.orient_by_gid) {
    flip.assign(nfac, 0);
    for (size_t f = nbfac; f < nfac; ++f)
      if (esuf[2 * f + 1] >= 0 && elem_gid[esuf[2 * f]] > elem_gid[esuf[2 * f + 1]]) flip[f] = 1;
  }
  auto left_of = [&](int f) { return (!flip.empty() && flip[f]) ? esuf[2 * f + 1] : esuf[2 * f]; };

  lap("face ids per (tet, local face)");
  // ---- BC type of every boundary face -------------------------------------
  // reference: bndSurfInt over the configured side sets of each BC type
  // (src/PDE/Integrate/Boundary.cpp:84-90); faces of unconfigured sets get no flux
  std::vector<int> bcface(nbfac, 0);
  if (bface && bface->nset > 0) {
    for (size_t s = 0; s < bface->nset; ++s) {
      int type = 0;
      for (size_t i = 0; i < ctx->bc_sideset.size(); ++i)
        if (ctx->bc_sideset[i] == bface->set_id[s]) {
          if (type != 0 && type != ctx->bc_type[i])
            return fail("qdg_mesh_upload: a side set is configured with two different BC types");
          type = ctx->bc_type[i];
        }
      if (type == 0) continue;
      for (size_t q = bface->set_off[s]; q < bface->set_off[s + 1]; ++q) {
        const size_t f = bface->face[q];
        if (f >= nbfac) return fail("qdg_mesh_upload: bface entry out of range");
        if (bcface[f] != 0) return fail("qdg_mesh_upload: a boundary face belongs to two configured side sets");
        bcface[f] = type;
      }
    }
  }

  // ---- device arrays -------------------------------------------------------
  const size_t stride = (ne + 63) / 64 * 64;
  std::vector<int> h_inpoel(4 * stride, 0), h_nbr(4 * stride, -1), h_finfo(4 * stride, 0), h_fid(4 * stride, 0);
  std::vector<double> h_vol(stride, 1.0);
  // device face numbering: by first touch in device element order (serial, cheap) ...
  std::vector<int> fmap(nfac, -1);
  int nfd = 0;
  for (size_t d = 0; d < nie; ++d) {
    const size_t h = d2h[d];
    for (int lf = 0; lf < 4; ++lf) {
      const int f = rface[4 * h + lf];
      if (fmap[f] < 0) fmap[f] = nfd++;
    }
  }
  // ... everything else per device row, on all cores
  std::atomic<int> bad_nbr{0};
  parallel_for(ne, [&](size_t d0, size_t d1) {
    for (size_t d = d0; d < d1; ++d) {
      const size_t h = d2h[d];
      for (int i = 0; i < 4; ++i) h_inpoel[i * stride + d] = nnew[inpoel[4 * h + i]];
      h_vol[d] = geoElem[4 * h];
      if (d >= nie) continue;
      for (int lf = 0; lf < 4; ++lf) {
        const int f = rface[4 * h + lf];
        h_fid[lf * stride + d] = fmap[f];
        const int nb = esuel[4 * h + lf];
        int info = ((size_t)left_of(f) == h) ? (1 << 6) : 0;
        if (nb < 0) {
          h_nbr[lf * stride + d] = -(1 + bcface[f]);
        } else {
          h_nbr[lf * stride + d] = h2d[nb];
          for (int j = 0; j < 3; ++j) {
            const size_t g = inpoel[4 * h + LPOFA[lf][j]];
            int m = -1;
            for (int q = 0; q < 4; ++q) if (inpoel[4 * (size_t)nb + q] == g) m = q;
            if (m < 0) { bad_nbr = 1; m = 0; }
            info |= m << (2 * j);
          }
        }
        h_finfo[lf * stride + d] = info;
      }
    }
  });
  if (bad_nbr) return fail("qdg_mesh_upload: neighbour does not share the face nodes (bad esuel/inpoel)");
  std::vector<double> h_area(std::max(nfd, 1)), h_nx(std::max(nfd, 1)), h_ny(std::max(nfd, 1)), h_nz(std::max(nfd, 1));
  for (size_t f = 0; f < nfac; ++f)
    if (fmap[f] >= 0) {
      const double sgn = (!flip.empty() && flip[f]) ? -1.0 : 1.0;   // the stored normal points out of the left tet
      h_area[fmap[f]] = geoFace[7 * f];
      h_nx[fmap[f]] = sgn * geoFace[7 * f + 1];
      h_ny[fmap[f]] = sgn * geoFace[7 * f + 2];
      h_nz[fmap[f]] = sgn * geoFace[7 * f + 3];
    }

  lap("device arrays (host side)");
  // ---- face tasks per tile (tile kernels of qdg_rhs_p1.hip) ---------------------
  // Tiles are runs of TILE consecutive device rows.  (Round 2 also measured runs cut at a task
  // count -- two full rounds of the workgroup's lanes: no faster; not kept.)
  std::vector<int> h_tile_row(1, 0);
  for (size_t r = TILE; r < nie; r += TILE) h_tile_row.push_back((int)r);
  h_tile_row.push_back((int)nie);
  const int ntile = (int)h_tile_row.size() - 1;
  int ntile_inner = 0;
  while (ntile_inner < ntile && (size_t)h_tile_row[ntile_inner + 1] <= ninner) ++ntile_inner;
  std::vector<int> h_tile_off(ntile + 1, 0), h_task_a, h_task_nb, h_task_f;
  int task_stride = 0;
  {
    struct Task { int key, a, nb, f; };
    // tasks of tile t in its (kind, local face) order; two passes over the tiles on all
    // cores: count, then fill at the tile's offset
    auto tile_tasks = [&](int t, std::vector<Task>& tt) {
      const size_t e0 = (size_t)h_tile_row[t], e1 = (size_t)h_tile_row[t + 1];
      tt.clear();
      for (size_t d = e0; d < e1; ++d)
        for (int lf = 0; lf < 4; ++lf) {
          const int nb = h_nbr[lf * stride + d], info = h_finfo[lf * stride + d];
          const int own_left = (info >> 6) & 1, code = info & 63;
          int kind, bc = 0, pl = 0, nbid = 0;
          if (nb < 0) { kind = TASK_BND; bc = -nb - 1; }
          else if ((size_t)nb >= e0 && (size_t)nb < e1) {
            if (!own_left) continue;            // listed by the face's left tet
            kind = TASK_INT; pl = nb - (int)e0;
          } else { kind = TASK_EXT; nbid = nb; }
          const int a = TASK_PACK(d - e0, lf, own_left, code, kind, bc, pl);
          // faces to other tiles first (their neighbour rows are requested at kernel entry),
          // then in-tile faces, then boundary faces
          const int rank = kind == TASK_EXT ? 0 : kind == TASK_INT ? 1 : 2;
          tt.push_back({ (rank << 2) | lf, a, nbid, h_fid[lf * stride + d] });
        }
      // same kind / local face next to each other: fewer divergent branches per wave
      std::stable_sort(tt.begin(), tt.end(), [](const Task& p, const Task& q) { return p.key < q.key; });
    };
    parallel_for((size_t)ntile, [&](size_t t0, size_t t1) {
      std::vector<Task> tt;
      for (size_t t = t0; t < t1; ++t) { tile_tasks((int)t, tt); h_tile_off[t + 1] = (int)tt.size(); }
    }, 256);
    for (int t = 0; t < ntile; ++t) h_tile_off[t + 1] += h_tile_off[t];
    const size_t ntask = (size_t)h_tile_off[ntile];
    // fixed-stride task lists for the uniform-order tile kernels (4 * TILE_BS slots per tile,
    // unused slots -1): no offset load in front of the descriptors, 1-2 % on the kernel (round 2:
    // 1.608 -> 1.581 ms at 10.1 M tets); p-adaptive runs (k_rhs_p1t) keep compact lists
    task_stride = !ctx->cfg.pref ? 4 * TILE_BS : 0;
    const size_t nslot = task_stride ? (size_t)ntile * task_stride : ntask;
    h_task_a.assign(nslot, -1); h_task_nb.assign(nslot, 0); h_task_f.assign(nslot, 0);
    parallel_for((size_t)ntile, [&](size_t t0, size_t t1) {
      std::vector<Task> tt;
      for (size_t t = t0; t < t1; ++t) {
        tile_tasks((int)t, tt);
        size_t o = task_stride ? t * (size_t)task_stride : (size_t)h_tile_off[t];
        for (const Task& k : tt) { h_task_a[o] = k.a; h_task_nb[o] = k.nb; h_task_f[o] = k.f; ++o; }
      }
    }, 256);
    lap("face tasks per tile");
    if (stats) {
      size_t cnt[3] = { 0, 0, 0 };
      for (int a : h_task_a) if (a >= 0) ++cnt[TASK_KIND(a)];
      std::fprintf(stderr, "qdg upload: %zu tets, %d tiles, tasks per tet: interior-in-tile %.3f, "
                   "to other tiles/ghosts %.3f, boundary %.3f; tasks per tile %.1f, rows per tile %.1f\n", nie, ntile,
                   (double)cnt[TASK_INT] / nie, (double)cnt[TASK_EXT] / nie, (double)cnt[TASK_BND] / nie,
                   (double)ntask / ntile, (double)nie / ntile);
    }
  }

  std::unique_ptr<qdg_mesh> m(new qdg_mesh);
  m->ctx = ctx;
  m->ndof = ctx->cfg.ndof;
  const int ncomp = ctx_ncomp(ctx);
  m->nprop = ncomp * m->ndof;
  m->nie = nie; m->ne = ne; m->stride = stride;
  hipStream_t s = ctx->stream;
  HIPCHK(m->inpoel.upload(h_inpoel, s));
  HIPCHK(m->nbr.upload(h_nbr, s));
  HIPCHK(m->finfo.upload(h_finfo, s));
  HIPCHK(m->fid.upload(h_fid, s));
  HIPCHK(m->d2h.upload(d2h, s));
  HIPCHK(m->x.upload(hx, s)); HIPCHK(m->y.upload(hy, s)); HIPCHK(m->z.upload(hz, s));
  HIPCHK(m->farea.upload(h_area, s)); HIPCHK(m->fnx.upload(h_nx, s));
  HIPCHK(m->fny.upload(h_ny, s)); HIPCHK(m->fnz.upload(h_nz, s));
  HIPCHK(m->vol.upload(h_vol, s));
  {
    std::vector<double> h_fgeo(4 * (size_t)std::max(nfd, 1)), h_xyz4(4 * (size_t)std::max(ncount, 1), 0.0);
    for (int i = 0; i < nfd; ++i) { h_fgeo[4*i] = h_area[i]; h_fgeo[4*i+1] = h_nx[i]; h_fgeo[4*i+2] = h_ny[i]; h_fgeo[4*i+3] = h_nz[i]; }
    for (int i = 0; i < ncount; ++i) { h_xyz4[4*i] = hx[i]; h_xyz4[4*i+1] = hy[i]; h_xyz4[4*i+2] = hz[i]; }
    HIPCHK(m->fgeo.upload(h_fgeo, s));
    HIPCHK(m->xyz4.upload(h_xyz4, s));
  }
  if (int rc = mesh_alloc_state(m.get(), ntile)) return rc;
  m->nnode_used = (size_t)ncount;

  DevMesh& dm = m->dm;
  dm.nie = (int)nie; dm.ne = (int)ne; dm.stride = (int)stride; dm.nnode = ncount; dm.nfac = nfd;
  dm.inpoel = m->inpoel.p; dm.nbr = m->nbr.p; dm.finfo = m->finfo.p; dm.fid = m->fid.p;
  dm.x = m->x.p; dm.y = m->y.p; dm.z = m->z.p;
  dm.farea = m->farea.p; dm.fnx = m->fnx.p; dm.fny = m->fny.p; dm.fnz = m->fnz.p;
  dm.vol = m->vol.p; dm.d2h = m->d2h.p;
  dm.fgeo = m->fgeo.p; dm.xyz4 = m->xyz4.p;
  HIPCHK(m->tile_row.upload(h_tile_row, s));
  HIPCHK(m->tile_off.upload(h_tile_off, s)); HIPCHK(m->task_a.upload(h_task_a, s));
  HIPCHK(m->task_nb.upload(h_task_nb, s)); HIPCHK(m->task_f.upload(h_task_f, s));
  dm.ntile = ntile; dm.ntile_inner = ntile_inner; dm.tile_row = m->tile_row.p; dm.task_stride = task_stride;
  dm.tile_rows = TILE;
  dm.tile_off = m->tile_off.p; dm.task_a = m->task_a.p;
  dm.task_nb = m->task_nb.p; dm.task_f = m->task_f.p;
  dm.tgeo = nullptr;
  if (task_stride > 0 && ctx->cfg.ndof == 4) {
    // face records in task order (DG-P1 tile kernel, version 2)
    HIPCHK(m->tgeo.alloc(4 * h_task_a.size()));
    launch_task_geo(h_task_a.size(), m->task_a.p, m->task_f.p, m->fgeo.p, m->tgeo.p, s);
    dm.tgeo = m->tgeo.p;
  }
  dm.blk0 = 0; dm.ninner = (int)ninner; dm.ncomp = ncomp; dm.pde = ctx->cfg.pde;
  dm.nlim = (int)nie;
  dm.ndofel = nullptr;
  if (ctx->cfg.pref) {
    HIPCHK(m->ndofel.alloc(ne)); HIPCHK(m->ndofel2.alloc(ne));
    launch_fill_int(m->ndofel.p, (int)ne, 4, s);
    dm.ndofel = m->ndofel.p;
  }
  HIPCHK(hipStreamSynchronize(s));
  lap("allocation + copies to the device");
  *out = m.release();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_mesh_destroy(qdg_mesh* mesh)
{
  QDG_TRY
  if (!mesh) return 0;
  (void)hipSetDevice(mesh->ctx->device);
  (void)hipStreamSynchronize(mesh->ctx->stream);
  qdg::StreamScope scope(mesh->ctx->stream);
  delete mesh;
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------- helpers

// the ghost rows of the current state, where a qdg_step_comm left them in another buffer (carry_pending)
static int flush_ghost_carry(qdg_mesh* mesh)
{
  double* src = mesh->carry_pending;
  mesh->carry_pending = nullptr;
  if (!src || src == mesh->Ucur || mesh->ne <= mesh->nie) return 0;
  HIPCHK(hipMemcpyAsync(mesh->Ucur + mesh->nie * (size_t)mesh->nprop, src + mesh->nie * (size_t)mesh->nprop,
                        (mesh->ne - mesh->nie) * (size_t)mesh->nprop * sizeof(double), hipMemcpyDeviceToDevice,
                        mesh->ctx->stream));
  return 0;
}

namespace qdg {
int mesh_flush_carry(qdg_mesh* mesh) { return mesh && mesh->carry_pending ? flush_ghost_carry(mesh) : 0; }
}  // namespace qdg

#define MESH_ENTER_(name, flush)                                 \
  if (!mesh) return fail(name ": null mesh");                    \
  qdg_ctx* ctx = mesh->ctx;                                      \
  HIPCHK(hipSetDevice(ctx->device));                             \
  hipStream_t s = ctx->stream;                                   \
  qdg::StreamScope stream_scope_(s);                             \
  if ((flush) && mesh->carry_pending) { if (int rcf_ = flush_ghost_carry(mesh)) return rcf_; } \
  mesh->slab_ready_for = nullptr;                                \
  (void)s
#define MESH_ENTER(name) MESH_ENTER_(name, true)
// (entry points that neither read nor move the state's ghost rows)
#define MESH_ENTER_NOFLUSH(name) MESH_ENTER_(name, false)

// host AoS (all ne rows) -> SoA planes `dst`
static int ensure_aos(qdg_mesh* mesh)
{
  if (!mesh->aos.p) HIPCHK(mesh->aos.alloc(mesh->ne * (size_t)mesh->nprop));
  return 0;
}

static int host_to_planes(qdg_mesh* mesh, const double* aos_host, double* dst)
{
  hipStream_t s = mesh->ctx->stream;
  const size_t n = mesh->ne * (size_t)mesh->nprop;
  if (int rc = ensure_aos(mesh)) return rc;
  HIPCHK(hipMemcpyAsync(mesh->aos.p, aos_host, n * sizeof(double), hipMemcpyHostToDevice, s));
  launch_aos2soa(mesh->aos.p, mesh->nprop, mesh->d2h.p, 0, (int)mesh->ne, (int)mesh->stride, dst, s);
  HIPCHK(hipGetLastError());
  return 0;
}

// SoA planes rows [0,nrows) -> host AoS; rows >= nrows of the host array are
// written as `fill_rest_zero ? 0 : unchanged`
static int planes_to_host(qdg_mesh* mesh, const double* src, size_t nrows, double* aos_host,
                          bool zero_rest)
{
  hipStream_t s = mesh->ctx->stream;
  const size_t n = mesh->ne * (size_t)mesh->nprop;
  if (int rc = ensure_aos(mesh)) return rc;
  if (zero_rest && nrows < mesh->ne)
    HIPCHK(hipMemsetAsync(mesh->aos.p, 0, n * sizeof(double), s));
  launch_soa2aos(src, mesh->nprop, mesh->d2h.p, 0, (int)nrows, (int)mesh->stride, mesh->aos.p, s);
  HIPCHK(hipGetLastError());
  if (zero_rest || nrows == mesh->ne) {
    HIPCHK(hipMemcpyAsync(aos_host, mesh->aos.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
  } else {
    // interior rows are not contiguous in the caller's numbering only if the
    // caller interleaves ghosts; by contract ghosts are rows [nie,ne)
    HIPCHK(hipMemcpyAsync(aos_host, mesh->aos.p, nrows * (size_t)mesh->nprop * sizeof(double),
                          hipMemcpyDeviceToHost, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

static double* free_buf(qdg_mesh* mesh, const double* a, const double* b)
{
  double* all[3] = { mesh->U.p, mesh->Un.p, mesh->W.p };
  for (double* p : all) if (p != a && p != b) return p;
  return nullptr;
}

static int scratch(qdg_mesh* mesh)
{
  const size_t fsz = (size_t)mesh->nprop * mesh->stride;
  if (!mesh->S1.p) HIPCHK(mesh->S1.alloc(fsz));
  if (!mesh->S2.p) HIPCHK(mesh->S2.alloc(fsz));
  return 0;
}

// owned_only: the stateless DGPDE-shaped call limits [0, nielem) as the reference does; the resident stages
// also limit the layer-1 ghosts of a chunk with two ghost layers (DevMesh::nlim, qdg_halo_set_depth)
static int run_limiter(qdg_mesh* mesh, double*& Ucur, double* Ualt_in, bool owned_only = false)
{
  double* Ualt = Ualt_in;
  qdg_ctx* ctx = mesh->ctx;
  hipStream_t s = ctx->stream;
  if (mesh->ndof == 1) return 0;          // DG.cpp:1251: rdof > 1 only
  DevMesh dm = mesh->dm;
  if (owned_only) dm.nlim = dm.nie;
  dm.lim_write_all = ctx->opt.limiter_write_all;
  if (ctx->cfg.limiter == QDG_LIMITER_SUPERBEEP1) {
    launch_superbee(mesh->ndof, dm, Ucur, s);
  } else if (ctx->cfg.limiter == QDG_LIMITER_WENOP1) {
    launch_weno(mesh->ndof, dm, ctx->ph.cweight, Ucur, Ualt, s);    // writes every row of Ualt
    std::swap(Ucur, Ualt);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// RHS dispatch: CompFlow DG-P1 has kernels of its own (qdg_rhs_p1.hip)
static bool use_p1_fast(const qdg_mesh* mesh)
{
  return mesh->ndof == 4 && mesh->dm.pde == QDG_PDE_COMPFLOW;
}

// tile / face-task kernel (each in-tile face evaluated once, LDS accumulation with
// ds_add_f64) unless bitwise run-to-run reproducibility is requested (option "p1_rhs" = 1)
static bool use_tile(const qdg_mesh* mesh)
{
  return mesh->ctx->opt.p1_rhs != 1 || mesh->dm.ndofel;      // p-adaptive DG exists in the tile kernel only
}

static void run_rhs(qdg_mesh* mesh, double t, const double* U, double* R)
{
  qdg_ctx* ctx = mesh->ctx;
  if (use_p1_fast(mesh) && use_tile(mesh))
    launch_rhs_p1t(mesh->dm, ctx->ph, t, U, R, false, mesh->blockmin.p, 1.0, DBL_MAX,
                   mesh->dtraw.p, mesh->dt_ptr, ctx->stream);
  else if (use_p1_fast(mesh))
    launch_rhs_p1(mesh->dm, ctx->ph, t, U, R, false, mesh->blockmin.p, 1.0, DBL_MAX,
                  mesh->dtraw.p, mesh->dt_ptr, ctx->stream);
  else
    launch_rhs(mesh->ndof, mesh->dm, ctx->ph, t, U, R, ctx->stream);
}

// ---------------------------------------------------------------- stateless

extern "C" int qdg_lhs(qdg_mesh* mesh, double* L_aos)
{
  QDG_TRY
  MESH_ENTER("qdg_lhs");
  if (!L_aos) return fail("qdg_lhs: null L");
  if (int rc = scratch(mesh)) return rc;
  launch_mass(mesh->ndof, mesh->dm, mesh->S1.p, s);
  HIPCHK(hipGetLastError());
  return planes_to_host(mesh, mesh->S1.p, mesh->ne, L_aos, false);
  QDG_CATCH
}

extern "C" int qdg_initialize(qdg_mesh* mesh, double t, double* U_aos)
{
  QDG_TRY
  MESH_ENTER("qdg_initialize");
  if (!U_aos) return fail("qdg_initialize: null U");
  if (int rc = scratch(mesh)) return rc;
  double* w = mesh->S1.p;
  launch_init(mesh->ndof, mesh->dm, ctx->ph, t, w, s);
  HIPCHK(hipGetLastError());
  return planes_to_host(mesh, w, mesh->nie, U_aos, false);
  QDG_CATCH
}

extern "C" int qdg_rhs(qdg_mesh* mesh, double t, const double* U_aos, double* R_aos)
{
  QDG_TRY
  MESH_ENTER("qdg_rhs");
  if (!U_aos || !R_aos) return fail("qdg_rhs: null U/R");
  if (int rc = scratch(mesh)) return rc;
  double* w = mesh->S1.p;
  if (int rc = host_to_planes(mesh, U_aos, w)) return rc;
  run_rhs(mesh, t, w, mesh->S2.p);
  HIPCHK(hipGetLastError());
  // ghost rows of R are returned as zero (the reference leaves partial sums
  // there that DG::solve never reads back: ghosts are overwritten by comsol)
  return planes_to_host(mesh, mesh->S2.p, mesh->nie, R_aos, true);
  QDG_CATCH
}

extern "C" int qdg_dt(qdg_mesh* mesh, const double* U_aos, double* mindt)
{
  QDG_TRY
  MESH_ENTER("qdg_dt");
  if (!U_aos || !mindt) return fail("qdg_dt: null argument");
  if (mesh->dm.pde == 1) {          // dg::Transport::dt: no estimate (DGTransport.hpp:189-199)
    *mindt = std::numeric_limits<double>::max();
    return 0;
  }
  if (int rc = scratch(mesh)) return rc;
  double* w = mesh->S1.p;
  if (int rc = host_to_planes(mesh, U_aos, w)) return rc;
  launch_dt(mesh->ndof, mesh->dm, ctx->ph, w, mesh->blockmin.p, 1.0, DBL_MAX, mesh->dtraw.p,
            mesh->diagout.p /*scratch*/, s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(mindt, mesh->dtraw.p, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_limit(qdg_mesh* mesh, double* U_aos)
{
  QDG_TRY
  MESH_ENTER("qdg_limit");
  if (!U_aos) return fail("qdg_limit: null U");
  // scratch pair that does not alias the resident state
  if (int rc = scratch(mesh)) return rc;
  double* a = mesh->S1.p;
  if (int rc = host_to_planes(mesh, U_aos, a)) return rc;
  if (int rc = run_limiter(mesh, a, mesh->S2.p, true)) return rc;
  return planes_to_host(mesh, a, mesh->ne, U_aos, false);
  QDG_CATCH
}

// WENO_P1 / Superbee_P1 as free functions of (esuel, U) -- what DG::lim calls before the
// first rhs/dt of a run (src/Inciter/DG.cpp:1251-1260; Limiter.cpp:29-316 read only esuel,
// ndofel and the solution): no mesh handle needed
extern "C" int qdg_limit_from(qdg_ctx* ctx, size_t nielem, size_t nunk, const int* esuel,
                              const size_t* ndofel, double* U_aos)
{
  QDG_TRY
  if (!ctx || !esuel || !U_aos) return fail("qdg_limit_from: null argument");
  if (nielem == 0 || nunk < nielem) return fail("qdg_limit_from: need 0 < nielem <= nunk");
  if (nunk > (size_t)(INT32_MAX - 64) / 64) return fail("qdg_limit_from: chunk too large for one call");
  if (ctx->cfg.ndof == 1 || ctx->cfg.limiter == QDG_LIMITER_NONE) return 0;    // DG.cpp:1251: rdof > 1 only
  if (ctx->cfg.pref && !ndofel) return fail("qdg_limit_from: p-adaptive DG needs ndofel");
  const size_t stride = (nunk + 63) / 64 * 64;
  std::vector<int> h(4 * stride, -1);
  for (size_t e = 0; e < nielem; ++e)
    for (int lf = 0; lf < 4; ++lf) {
      const int nb = esuel[4 * e + lf];
      if (nb < -1 || (nb >= 0 && (size_t)nb >= nunk)) return fail("qdg_limit_from: esuel entry out of range");
      h[lf * stride + e] = nb;
    }
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int ncomp = ctx_ncomp(ctx);
  const size_t np = (size_t)ncomp * ctx->cfg.ndof, fsz = np * stride;
  DevBuf<int> dn, dnd;
  DevBuf<double> d;
  HIPCHK(dn.upload(h, s));
  HIPCHK(d.alloc(2 * fsz));
  HIPCHK(hipMemsetAsync(d.p, 0, 2 * fsz * 8, s));
  HIPCHK(hipMemcpyAsync(d.p, U_aos, np * nunk * 8, hipMemcpyHostToDevice, s));
  DevMesh dm{};
  dm.nie = (int)nielem; dm.ne = (int)nunk; dm.stride = (int)stride; dm.ncomp = ncomp; dm.pde = ctx->cfg.pde; dm.nbr = dn.p;
  if (ctx->cfg.pref) {
    std::vector<int> nd(nunk);
    for (size_t e = 0; e < nunk; ++e) nd[e] = (int)ndofel[e];
    HIPCHK(dnd.upload(nd, s));
    dm.ndofel = dnd.p;
  }
  double* cur = d.p;
  if (ctx->cfg.limiter == QDG_LIMITER_SUPERBEEP1) {
    launch_superbee(ctx->cfg.ndof, dm, cur, s);
  } else {
    launch_weno(ctx->cfg.ndof, dm, ctx->ph.cweight, cur, d.p + fsz, s);
    cur = d.p + fsz;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(U_aos, cur, np * nunk * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------- resident

extern "C" int qdg_state_upload(qdg_mesh* mesh, const double* U_aos)
{
  QDG_TRY
  MESH_ENTER("qdg_state_upload");
  if (!U_aos) return fail("qdg_state_upload: null U");
  if (int rc = host_to_planes(mesh, U_aos, mesh->Ucur)) return rc;
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_state_download(qdg_mesh* mesh, double* U_aos)
{
  QDG_TRY
  MESH_ENTER("qdg_state_download");
  if (!U_aos) return fail("qdg_state_download: null U");
  return planes_to_host(mesh, mesh->Ucur, mesh->ne, U_aos, false);
  QDG_CATCH
}

extern "C" int qdg_state_initialize(qdg_mesh* mesh, double t)
{
  QDG_TRY
  MESH_ENTER("qdg_state_initialize");
  launch_init(mesh->ndof, mesh->dm, ctx->ph, t, mesh->Ucur, s);
  if (mesh->dm.ndofel) launch_fill_int(mesh->ndofel.p, (int)mesh->ne, mesh->ndof, s);   // DG.cpp:927
  HIPCHK(hipGetLastError());
  return 0;
  QDG_CATCH
}

// Problem::fieldNames (src/PDE/CompFlow/Problem/*.cpp; dg::Transport::fieldNames,
// DGTransport.hpp:211-229 with depvar 'c')
static const char* problem_field_name(const qdg_ctx* ctx, size_t f)
{
  static const char* six[] = { "density_numerical", "x-velocity_numerical", "y-velocity_numerical",
                               "z-velocity_numerical", "specific_total_energy_numerical",
                               "pressure_numerical" };
  static const char* vort[] = { "density_numerical", "density_analytical", "x-velocity_numerical",
                                "x-velocity_analytical", "y-velocity_numerical", "y-velocity_analytical",
                                "z-velocity_numerical", "z-velocity_analytical",
                                "specific_total_energy_numerical", "specific_total_energy_analytical",
                                "pressure_numerical", "pressure_analytical" };
  static const char* tg[] = { "density_numerical", "density_analytical", "x-velocity_numerical",
                              "x-velocity_analytical", "err(u)", "y-velocity_numerical",
                              "y-velocity_analytical", "err(v)", "z-velocity_numerical",
                              "z-velocity_analytical", "specific_total_energy_numerical",
                              "specific_total_energy_analytical", "err(E)", "pressure_numerical",
                              "pressure_analytical" };
  static const char* ms[] = { "density_numerical", "x-velocity_numerical", "y-velocity_numerical",
                              "z-velocity_numerical", "specific_total_energy_numerical",
                              "pressure_numerical", "density_analytical", "x-velocity_analytical",
                              "y-velocity_analytical", "z-velocity_analytical",
                              "specific_total_energy_analytical", "pressure_analytical", "err(rho)",
                              "err(e)", "err(p)", "err(u)", "err(v)", "err(w)" };
  static const char* ud[] = { "density", "x-velocity", "y-velocity", "z-velocity",
                              "specific total energy", "pressure", "temperature" };
  // depvar 'c' + component + suffix, the three blocks of DGTransport.hpp:211-228
  static const char* trn[] = { "c0_numerical", "c1_numerical", "c2_numerical", "c3_numerical", "c4_numerical",
                               "c0_analytic", "c1_analytic", "c2_analytic", "c3_analytic", "c4_analytic",
                               "c0_error", "c1_error", "c2_error", "c3_error", "c4_error" };
  const int ncomp = ctx_ncomp(ctx);
  const size_t n = (size_t)field_count(ctx->cfg.pde, ncomp, ctx->cfg.problem);
  if (f >= n) return (f == n && ctx->cfg.pref) ? "ndof" : "";
  if (ctx->cfg.pde == QDG_PDE_TRANSPORT) return trn[5 * (f / ncomp) + f % ncomp];
  switch (ctx->cfg.problem) {
    case QDG_PROBLEM_VORTICAL_FLOW: return vort[f];
    case QDG_PROBLEM_TAYLOR_GREEN: return tg[f];
    case QDG_PROBLEM_NL_ENERGY_GROWTH: case QDG_PROBLEM_RAYLEIGH_TAYLOR: return ms[f];
    case QDG_PROBLEM_USER_DEFINED: return ud[f];
    default: return six[f];
  }
}

extern "C" int qdg_ctx_field_count(qdg_ctx* ctx, size_t* nfield)
{
  QDG_TRY
  if (!ctx || !nfield) return fail("qdg_ctx_field_count: null argument");
  *nfield = (size_t)field_count(ctx->cfg.pde, ctx_ncomp(ctx), ctx->cfg.problem) +
            (ctx->cfg.pref ? 1 : 0);
  return 0;
  QDG_CATCH
}

extern "C" const char* qdg_ctx_field_name(qdg_ctx* ctx, size_t f)
{
  return ctx ? problem_field_name(ctx, f) : "";
}

extern "C" int qdg_field_count(qdg_mesh* mesh, size_t* nfield)
{
  QDG_TRY
  if (!mesh || !nfield) return fail("qdg_field_count: null argument");
  return qdg_ctx_field_count(mesh->ctx, nfield);
  QDG_CATCH
}

extern "C" const char* qdg_field_name(qdg_mesh* mesh, size_t f)
{
  return mesh ? problem_field_name(mesh->ctx, f) : "";
}

extern "C" int qdg_field_output(qdg_mesh* mesh, double t, double* out)
{
  QDG_TRY
  MESH_ENTER("qdg_field_output");
  if (!out) return fail("qdg_field_output: null out");
  size_t nf = 0;
  if (int rc = qdg_ctx_field_count(ctx, &nf)) return rc;
  const size_t n = nf * mesh->nie;
  if (mesh->fout.n < n) HIPCHK(mesh->fout.alloc(n));
  launch_field_output(mesh->ndof, mesh->dm, ctx->ph, t, mesh->Ucur, nullptr, (int)mesh->nie, mesh->fout.p, s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, mesh->fout.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

// ---- stateless output-side members of the DGPDE interface (no mesh handle) -------------

extern "C" int qdg_field_output_from(qdg_ctx* ctx, double t, size_t nunk, const double* geoElem,
                                     const double* U_aos, double* out)
{
  QDG_TRY
  if (!ctx || (nunk && (!geoElem || !U_aos || !out))) return fail("qdg_field_output_from: null argument");
  if (nunk > (size_t)INT32_MAX / 64) return fail("qdg_field_output_from: too many elements for one call");
  if (nunk == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int ncomp = ctx_ncomp(ctx);
  const size_t np = (size_t)ncomp * ctx->cfg.rdof, nf = (size_t)field_count(ctx->cfg.pde, ncomp, ctx->cfg.problem);
  DevBuf<double> d;
  HIPCHK(d.alloc((np + 4 + nf) * nunk));
  double* dU = d.p; double* dG = d.p + np * nunk; double* dO = dG + 4 * nunk;
  HIPCHK(hipMemcpyAsync(dU, U_aos, np * nunk * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dG, geoElem, 4 * nunk * 8, hipMemcpyHostToDevice, s));
  DevMesh dm{};
  dm.ncomp = ncomp; dm.pde = ctx->cfg.pde;
  launch_field_output(ctx->cfg.rdof, dm, ctx->ph, t, dU, dG, (int)nunk, dO, s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, dO, nf * nunk * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_avg_elem_to_node(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                                    const double* U_aos, double* out)
{
  QDG_TRY
  if (!ctx || !inpoel || !U_aos || !out) return fail("qdg_avg_elem_to_node: null argument");
  if (ctx->cfg.pde != QDG_PDE_COMPFLOW)
    return fail("qdg_avg_elem_to_node: dg::Transport returns no nodal fields (DGTransport.hpp:231-239)");
  if (nelem > (size_t)INT32_MAX / 64 || nnode > (size_t)INT32_MAX / 8)
    return fail("qdg_avg_elem_to_node: mesh too large for one call");
  std::vector<int> h(4 * nelem);
  for (size_t i = 0; i < 4 * nelem; ++i) {
    if (inpoel[i] >= nnode) return fail("qdg_avg_elem_to_node: inpoel entry out of range");
    h[i] = (int)inpoel[i];
  }
  if (nnode == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const size_t np = (size_t)NCOMP * ctx->cfg.rdof;
  DevBuf<int> di;
  DevBuf<double> d;
  HIPCHK(di.upload(h, s));
  HIPCHK(d.alloc(np * nelem + 7 * nnode));
  double* dU = d.p; double* dO = d.p + np * nelem; double* dC = dO + 6 * nnode;
  HIPCHK(hipMemcpyAsync(dU, U_aos, np * nelem * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(dO, 0, 7 * nnode * 8, s));
  launch_avg_elem_to_node(ctx->ph, ctx->cfg.rdof, (int)nelem, (int)nnode, di.p, dU, dO, dC, s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, dO, 6 * nnode * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_initialize_from(qdg_ctx* ctx, size_t nielem, size_t nnode, const size_t* inpoel,
                                   const double* x, const double* y, const double* z,
                                   const double* L_aos, double t, double* U_aos)
{
  QDG_TRY
  if (!ctx || !inpoel || !x || !y || !z || !U_aos) return fail("qdg_initialize_from: null argument");
  if (nielem > (size_t)(INT32_MAX - 64) / 64 || nnode > (size_t)INT32_MAX)
    return fail("qdg_initialize_from: mesh too large for one call");
  if (nielem == 0) return 0;
  const size_t stride = (nielem + 63) / 64 * 64;
  std::vector<int> h(4 * stride, 0);
  for (size_t e = 0; e < nielem; ++e)
    for (int i = 0; i < 4; ++i) {
      if (inpoel[4 * e + i] >= nnode) return fail("qdg_initialize_from: inpoel entry out of range");
      h[i * stride + e] = (int)inpoel[4 * e + i];
    }
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int ncomp = ctx_ncomp(ctx);
  const size_t np = (size_t)ncomp * ctx->cfg.ndof;
  DevBuf<int> di;
  DevBuf<double> d;
  HIPCHK(di.upload(h, s));
  HIPCHK(d.alloc(3 * nnode + stride + np * nielem));
  double* dx = d.p; double* dy = dx + nnode; double* dz = dy + nnode; double* dv = dz + nnode;
  double* dU = dv + stride;
  HIPCHK(hipMemcpyAsync(dx, x, nnode * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dy, y, nnode * 8, hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dz, z, nnode * 8, hipMemcpyHostToDevice, s));
  if (L_aos) {
    // the caller's mass matrix: L(e, 0) = vol_e (Mass.cpp:25-73), as tk::initialize divides by it
    std::vector<double> hv(stride, 1.0);
    for (size_t e = 0; e < nielem; ++e) hv[e] = L_aos[e * np];
    HIPCHK(hipMemcpyAsync(dv, hv.data(), stride * 8, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
  } else {
    launch_tet_volumes((int)nielem, (int)stride, di.p, dx, dy, dz, dv, s);
  }
  DevMesh dm{};
  dm.nie = dm.ne = (int)nielem; dm.stride = (int)stride; dm.nnode = (int)nnode; dm.ncomp = ncomp; dm.pde = ctx->cfg.pde;
  dm.inpoel = di.p; dm.x = dx; dm.y = dy; dm.z = dz; dm.vol = dv;
  launch_init(ctx->cfg.ndof, dm, ctx->ph, t, dU, s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(U_aos, dU, np * nielem * 8, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_state_device_ptr(qdg_mesh* mesh, void** dptr, size_t* stride)
{
  QDG_TRY
  if (!mesh || !dptr || !stride) return fail("qdg_state_device_ptr: null argument");
  if (mesh->carry_pending) {
    HIPCHK(hipSetDevice(mesh->ctx->device));
    if (int rc = flush_ghost_carry(mesh)) return rc;
  }
  *dptr = mesh->Ucur; *stride = mesh->stride;
  mesh->slab_ready_for = nullptr;      // the caller may write the state through this pointer
  return 0;
  QDG_CATCH
}

extern "C" int qdg_stage_limit(qdg_mesh* mesh)
{
  QDG_TRY
  MESH_ENTER("qdg_stage_limit");
  return run_limiter(mesh, mesh->Ucur, free_buf(mesh, mesh->Ucur, mesh->Unp));
  QDG_CATCH
}

extern "C" int qdg_stage_dt(qdg_mesh* mesh, double tleft)
{
  QDG_TRY
  MESH_ENTER("qdg_stage_dt");
  if (ctx->cfg.dt > 0.0) {
    // constant dt configured (DG.cpp:1386-1393)
    const double v = std::min(ctx->cfg.dt, tleft);
    // zero blocks to reduce: the final kernel writes min(DBL_MAX*1, v) = v
    DevMesh none{};
    launch_dt(mesh->ndof, none, ctx->ph, nullptr, mesh->blockmin.p, 1.0, v, mesh->dtraw.p,
              mesh->dt_ptr, s);
    HIPCHK(hipGetLastError());
    return 0;
  }
  const double p = (mesh->ndof == 4) ? 1.0 : (mesh->ndof == 10) ? 2.0 : 0.0;
  const double scale = ctx->cfg.cfl / (2.0 * p + 1.0);     // DG.cpp:1404-1418
  launch_dt(mesh->ndof, mesh->dm, ctx->ph, mesh->Ucur, mesh->blockmin.p, scale, tleft,
            mesh->dtraw.p, mesh->dt_ptr, s);
  HIPCHK(hipGetLastError());
  return 0;
  QDG_CATCH
}

extern "C" int qdg_stage_dt_get(qdg_mesh* mesh, double* dt_host)
{
  QDG_TRY
  MESH_ENTER_NOFLUSH("qdg_stage_dt_get");
  if (!dt_host) return fail("qdg_stage_dt_get: null argument");
  HIPCHK(hipMemcpyAsync(dt_host, mesh->dt_ptr, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_stage_dt_set(qdg_mesh* mesh, double dt)
{
  QDG_TRY
  MESH_ENTER("qdg_stage_dt_set");
  HIPCHK(hipMemcpyAsync(mesh->dt_ptr, &dt, sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

extern "C" int qdg_stage_dt_device_ptr(qdg_mesh* mesh, void** dptr)
{
  QDG_TRY
  if (!mesh || !dptr) return fail("qdg_stage_dt_device_ptr: null argument");
  *dptr = mesh->dt_ptr;
  return 0;
  QDG_CATCH
}

static const double RK[2][3] = { { 0.0, 3.0 / 4.0, 1.0 / 3.0 }, { 1.0, 1.0 / 4.0, 2.0 / 3.0 } };   // DG.cpp:39-40

static int prof_begin(qdg_mesh* mesh, std::pair<hipEvent_t, hipEvent_t>** ev, bool cont = false, int kind = 0)
{
  *ev = nullptr;
  if (!mesh->prof) return 0;
  if (mesh->ev_used == mesh->ev.size()) {
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    mesh->ev.emplace_back(a, b);
    mesh->ev_cont.push_back(0);
    mesh->ev_kind.push_back(0);
  }
  mesh->ev_cont[mesh->ev_used] = cont ? 1 : 0;
  mesh->ev_kind[mesh->ev_used] = (char)kind;
  *ev = &mesh->ev[mesh->ev_used++];
  HIPCHK(hipEventRecord((*ev)->first, mesh->ctx->stream));
  return 0;
}

// rhs + update with dt already in the device scalar (in-place form; Un is a copy)
extern "C" int qdg_stage_rhs_update(qdg_mesh* mesh, int stage, double t)
{
  QDG_TRY
  MESH_ENTER("qdg_stage_rhs_update");
  if (stage < 0 || stage > 2) return fail("qdg_stage_rhs_update: stage must be 0,1,2");
  const size_t fsz = (size_t)mesh->nprop * mesh->stride * sizeof(double);
  if (stage == 0) {   // m_un = m_u, DG.cpp:1472
    mesh->Unp = free_buf(mesh, mesh->Ucur, nullptr);
    HIPCHK(hipMemcpyAsync(mesh->Unp, mesh->Ucur, fsz, hipMemcpyDeviceToDevice, s));
  }
  if (!mesh->Unp || mesh->Unp == mesh->Ucur) return fail("qdg_stage_rhs_update: call stage 0 first");
  std::pair<hipEvent_t, hipEvent_t>* ev;
  if (int rc = prof_begin(mesh, &ev)) return rc;
  run_rhs(mesh, t, mesh->Ucur, mesh->R.p);
  if (ev) HIPCHK(hipEventRecord(ev->second, s));
  launch_rk(mesh->ndof, mesh->dm, RK[0][stage], RK[1][stage], mesh->dt_ptr, mesh->Unp, mesh->R.p,
            mesh->Ucur, mesh->Ucur, s);
  HIPCHK(hipGetLastError());
  if (stage == 2) mesh->Unp = nullptr;
  return 0;
  QDG_CATCH
}

static bool can_fuse_update_limit(const qdg_mesh* mesh);

// Fused form.  Stage 0: Un is the current buffer itself (no copy: the update
// writes the new state to another buffer), R <- rhs(U) and, with a CFL time
// step, dt from the same kernel.  Stages 1, 2 on the DG-P1 fast path: the RK
// update is fused into the RHS kernel (dt is known), R never goes to memory.
extern "C" int qdg_stage_rhs_dt(qdg_mesh* mesh, int stage, double t, double tleft)
{
  QDG_TRY
  MESH_ENTER("qdg_stage_rhs_dt");
  if (stage < 0 || stage > 2) return fail("qdg_stage_rhs_dt: stage must be 0,1,2");
  if (stage == 0) mesh->Unp = mesh->Ucur;           // m_un = m_u, DG.cpp:1472
  if (!mesh->Unp) return fail("qdg_stage_rhs_dt: call stage 0 first");
  mesh->Upending = nullptr;
  const bool cfl_dt = stage == 0 && !(ctx->cfg.dt > 0.0);
  if (stage == 0 && !cfl_dt) if (int rc = qdg_stage_dt(mesh, tleft)) return rc;
  std::pair<hipEvent_t, hipEvent_t>* ev;
  // the update is fused into the RHS kernel whenever dt is known before the launch: stages 1, 2
  // always, stage 0 when the time step is prescribed (Un is then the current buffer itself and
  // a = 0, b = 1)
  // (not when the stage-0 update is going to be fused with the limiter of stage 1 instead)
  const bool fuse = stage > 0 || (!cfl_dt && !can_fuse_update_limit(mesh));
  if (use_p1_fast(mesh) && fuse) {
    double* out = free_buf(mesh, mesh->Ucur, mesh->Unp);
    if (int rc = prof_begin(mesh, &ev)) return rc;
    if (use_tile(mesh))
      launch_rhs_p1t_rk(mesh->dm, ctx->ph, t, mesh->Ucur, out, RK[0][stage], RK[1][stage],
                        mesh->dt_ptr, mesh->Unp, s);
    else
      launch_rhs_p1_rk(mesh->dm, ctx->ph, t, mesh->Ucur, out, RK[0][stage], RK[1][stage],
                       mesh->dt_ptr, mesh->Unp, s);
    if (ev) HIPCHK(hipEventRecord(ev->second, s));
    mesh->Upending = out;
  } else if (fuse && mesh->dm.pde == QDG_PDE_COMPFLOW) {
    // P0 / P2 (and the generic P1 path): the same fusion in the generic kernel
    double* out = free_buf(mesh, mesh->Ucur, mesh->Unp);
    if (int rc = prof_begin(mesh, &ev)) return rc;
    launch_rhs_rk(mesh->ndof, mesh->dm, ctx->ph, t, mesh->Ucur, out, RK[0][stage], RK[1][stage],
                  mesh->dt_ptr, mesh->Unp, s);
    if (ev) HIPCHK(hipEventRecord(ev->second, s));
    mesh->Upending = out;
  } else if (cfl_dt && use_p1_fast(mesh)) {
    const double scale = ctx->cfg.cfl / 3.0;     // cfl/(2p+1), p = 1 (DG.cpp:1404-1418)
    if (int rc = prof_begin(mesh, &ev)) return rc;
    // here the event pair also covers the 1-block dt reduction (~5 us)
    if (use_tile(mesh))
      launch_rhs_p1t(mesh->dm, ctx->ph, t, mesh->Ucur, mesh->R.p, true, mesh->blockmin.p, scale,
                     tleft, mesh->dtraw.p, mesh->dt_ptr, s);
    else
      launch_rhs_p1(mesh->dm, ctx->ph, t, mesh->Ucur, mesh->R.p, true, mesh->blockmin.p, scale,
                    tleft, mesh->dtraw.p, mesh->dt_ptr, s);
    if (ev) HIPCHK(hipEventRecord(ev->second, s));
  } else if (cfl_dt && mesh->dm.pde == QDG_PDE_COMPFLOW) {
    // P0 / P2 (and the generic P1 path): the CFL sum comes out of the same face loop
    const double p = (mesh->ndof == 4) ? 1.0 : (mesh->ndof == 10) ? 2.0 : 0.0;
    const double scale = ctx->cfg.cfl / (2.0 * p + 1.0);     // DG.cpp:1404-1418
    if (int rc = prof_begin(mesh, &ev)) return rc;
    launch_rhs_dt(mesh->ndof, mesh->dm, ctx->ph, t, mesh->Ucur, mesh->R.p, mesh->blockmin.p, scale,
                  tleft, mesh->dtraw.p, mesh->dt_ptr, s);
    if (ev) HIPCHK(hipEventRecord(ev->second, s));
  } else {
    if (cfl_dt) if (int rc = qdg_stage_dt(mesh, tleft)) return rc;
    if (int rc = prof_begin(mesh, &ev)) return rc;
    run_rhs(mesh, t, mesh->Ucur, mesh->R.p);
    if (ev) HIPCHK(hipEventRecord(ev->second, s));
  }
  HIPCHK(hipGetLastError());
  return 0;
  QDG_CATCH
}

extern "C" int qdg_stage_update(qdg_mesh* mesh, int stage)
{
  QDG_TRY
  MESH_ENTER("qdg_stage_update");
  if (stage < 0 || stage > 2) return fail("qdg_stage_update: stage must be 0,1,2");
  if (!mesh->Unp) return fail("qdg_stage_update: no stage in flight");
  if (mesh->Upending) {                       // the fused kernel already wrote the new state
    // ghost rows are not written by the update: carry them over, as the unfused path does --
    // except inside qdg_step_comm before a stage whose first action is to receive them again
    if (mesh->ne > mesh->nie && !mesh->skip_ghost_carry)
      HIPCHK(hipMemcpyAsync(mesh->Upending + mesh->nie * (size_t)mesh->nprop,
                            mesh->Ucur + mesh->nie * (size_t)mesh->nprop,
                            (mesh->ne - mesh->nie) * (size_t)mesh->nprop * sizeof(double),
                            hipMemcpyDeviceToDevice, s));
    else if (mesh->ne > mesh->nie)
      mesh->carry_src = mesh->Ucur;           // where the ghost rows still are, should the receive not happen
    mesh->Ucur = mesh->Upending;
    mesh->Upending = nullptr;
  } else {
    double* out = free_buf(mesh, mesh->Ucur, mesh->Unp);
    launch_rk(mesh->ndof, mesh->dm, RK[0][stage], RK[1][stage], mesh->dt_ptr, mesh->Unp, mesh->R.p,
              mesh->Ucur, out, s);
    HIPCHK(hipGetLastError());
    // ghost rows are not touched by the update; carry them over so that a
    // caller reading ghosts before the next exchange sees the old values
    if (mesh->ne > mesh->nie)
      HIPCHK(hipMemcpyAsync(out + mesh->nie * (size_t)mesh->nprop, mesh->Ucur + mesh->nie * (size_t)mesh->nprop,
                            (mesh->ne - mesh->nie) * (size_t)mesh->nprop * sizeof(double),
                            hipMemcpyDeviceToDevice, s));
    mesh->Ucur = out;
  }
  if (stage == 2) mesh->Unp = nullptr;
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------- p-adaptive DG

extern "C" int qdg_stage_pdg_eval(qdg_mesh* mesh)
{
  QDG_TRY
  MESH_ENTER("qdg_stage_pdg_eval");
  if (!mesh->dm.ndofel) return fail("qdg_stage_pdg_eval: the context was not created with pref");
  launch_pdg_eval(mesh->dm, mesh->Ucur, ctx->cfg.tolref, mesh->ndofel.p, s);        // DG::next
  HIPCHK(hipGetLastError());
  return 0;
  QDG_CATCH
}

extern "C" int qdg_stage_pdg_propagate(qdg_mesh* mesh)
{
  QDG_TRY
  MESH_ENTER("qdg_stage_pdg_propagate");
  if (!mesh->dm.ndofel) return fail("qdg_stage_pdg_propagate: the context was not created with pref");
  launch_pdg_propagate(mesh->dm, mesh->ndofel.p, mesh->ndofel2.p, s);               // DG::lim
  HIPCHK(hipMemcpyAsync(mesh->ndofel.p, mesh->ndofel2.p, mesh->ne * sizeof(int),
                        hipMemcpyDeviceToDevice, s));
  launch_pdg_zero(mesh->dm, mesh->ndofel.p, mesh->Ucur, s);                          // DG::solve
  HIPCHK(hipGetLastError());
  return 0;
  QDG_CATCH
}

extern "C" int qdg_stage_pdg(qdg_mesh* mesh)
{
  QDG_TRY
  if (!mesh) return fail("qdg_stage_pdg: null mesh");
  if (mesh->nnbr != 0)
    return fail("qdg_stage_pdg: this chunk has halo neighbours; call qdg_stage_pdg_eval, exchange "
                "the ghosts, then qdg_stage_pdg_propagate");
  if (int rc = qdg_stage_pdg_eval(mesh)) return rc;
  return qdg_stage_pdg_propagate(mesh);
  QDG_CATCH
}

extern "C" int qdg_ndofel_get(qdg_mesh* mesh, size_t* ndofel)
{
  QDG_TRY
  MESH_ENTER("qdg_ndofel_get");
  if (!ndofel) return fail("qdg_ndofel_get: null argument");
  if (!mesh->dm.ndofel) {
    for (size_t e = 0; e < mesh->ne; ++e) ndofel[e] = (size_t)mesh->ndof;
    return 0;
  }
  std::vector<int> nd(mesh->ne), d2h(mesh->ne);
  HIPCHK(hipMemcpyAsync(nd.data(), mesh->ndofel.p, mesh->ne * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipMemcpyAsync(d2h.data(), mesh->d2h.p, mesh->ne * sizeof(int), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  for (size_t d = 0; d < mesh->ne; ++d) ndofel[d2h[d]] = (size_t)nd[d];
  return 0;
  QDG_CATCH
}

extern "C" int qdg_ndofel_set(qdg_mesh* mesh, const size_t* ndofel)
{
  QDG_TRY
  MESH_ENTER("qdg_ndofel_set");
  if (!ndofel) return fail("qdg_ndofel_set: null argument");
  if (!mesh->dm.ndofel) return fail("qdg_ndofel_set: the context was not created with pref");
  std::vector<int> nd(mesh->ne), d2h(mesh->ne);
  HIPCHK(hipMemcpy(d2h.data(), mesh->d2h.p, mesh->ne * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t d = 0; d < mesh->ne; ++d) {
    const size_t v = ndofel[d2h[d]];
    if (v != 1 && v != 4) return fail("qdg_ndofel_set: entries must be 1 or 4");
    nd[d] = (int)v;
  }
  HIPCHK(hipMemcpyAsync(mesh->ndofel.p, nd.data(), mesh->ne * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

// Stage-0 update fused with the limiter of stage 1 (k_upd_superbee): DG-P1 CompFlow with
// Superbee, uniform order.  Option "fused_update" = 0 keeps the two kernels (A/B runs).
static bool can_fuse_update_limit(const qdg_mesh* mesh)
{
  const bool off = !mesh->ctx->opt.fused_update;
  return !off && use_p1_fast(mesh) && mesh->ctx->cfg.limiter == QDG_LIMITER_SUPERBEEP1 && !mesh->dm.ndofel;
}

// after qdg_stage_rhs_dt(stage 0): U1 = Superbee(U0 + dt R / L) into a free buffer; the
// ghost rows of that buffer must already hold the neighbours' U1 (exchange_upd)
static int stage0_update_and_limit(qdg_mesh* mesh)
{
  qdg_ctx* ctx = mesh->ctx;
  hipStream_t s = ctx->stream;
  if (!mesh->Unp || mesh->Unp != mesh->Ucur || mesh->Upending)
    return fail("stage0_update_and_limit: not after a stage-0 RHS");
  double* out = free_buf(mesh, mesh->Ucur, mesh->Unp);
  launch_upd_superbee(mesh->dm, mesh->dt_ptr, mesh->Ucur, mesh->R.p, out, s);
  HIPCHK(hipGetLastError());
  mesh->Ucur = out;                   // Unp keeps pointing at the stage-0 state
  return 0;
}

extern "C" int qdg_step(qdg_mesh* mesh, double t, double tleft, double* dt_taken)
{
  QDG_TRY
  if (!mesh) return fail("qdg_step: null mesh");
  if (mesh->nnbr != 0)
    return fail("qdg_step: this chunk has halo neighbours; drive the stages and the exchange "
                "explicitly (qdg_stage_* + qdg_halo_*)");
  const bool fuse = can_fuse_update_limit(mesh);
  for (int stage = 0; stage < 3; ++stage) {
    if (stage == 0 && mesh->dm.ndofel) if (int rc = qdg_stage_pdg(mesh)) return rc;
    if (!(fuse && stage == 1)) if (int rc = qdg_stage_limit(mesh)) return rc;   // stage 1: done below
    if (int rc = qdg_stage_rhs_dt(mesh, stage, t, tleft)) return rc;
    if (fuse && stage == 0) { if (int rc = stage0_update_and_limit(mesh)) return rc; }
    else if (int rc = qdg_stage_update(mesh, stage)) return rc;
  }
  if (dt_taken) return qdg_stage_dt_get(mesh, dt_taken);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_diag(qdg_mesh* mesh, double t_new, double* out15)
{
  QDG_TRY
  MESH_ENTER("qdg_diag");
  if (!out15) return fail("qdg_diag: null out");
  launch_diag(mesh->ndof, mesh->dm, ctx->ph, t_new, mesh->Ucur, mesh->diagpart.p, mesh->diagout.p, s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out15, mesh->diagout.p, 15 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------- state transfer after AMR

extern "C" int qdg_state_transfer(qdg_mesh* from, qdg_mesh* to, const size_t* parent_of_child)
{
  QDG_TRY
  if (!from || !to || !parent_of_child) return fail("qdg_state_transfer: null argument");
  if (from->ctx != to->ctx) return fail("qdg_state_transfer: the two meshes must belong to one context");
  if (from->nprop != to->nprop) return fail("qdg_state_transfer: row length differs");
  if (from->dm.ndofel || to->dm.ndofel)
    return fail("qdg_state_transfer: p-adaptive runs are not combined with mesh refinement "
                "(DG::resizePostAMR does not carry m_ndof over either)");
  // (the parent list is converted and applied on the device: no host loop over the tets)
  return qdg::dev_state_transfer(from, to, parent_of_child);
  QDG_CATCH
}

// State migration after a re-partition (DG::resizePostAMR + the load balancing that follows it,
// src/Inciter/DG.cpp:1537-1664): rows travel between chunks by GLOBAL tet id.
extern "C" int qdg_state_migrate(qdg_mesh* from, const size_t* from_gid, qdg_mesh* to, const size_t* to_gid,
                                 size_t* nmoved)
{
  QDG_TRY
  if (!from || !to || !from_gid || !to_gid) return fail("qdg_state_migrate: null argument");
  if (from->ctx != to->ctx) return fail("qdg_state_migrate: the two meshes must belong to one context");
  if (from->nprop != to->nprop) return fail("qdg_state_migrate: row length differs");
  if (from->dm.ndofel || to->dm.ndofel) return fail("qdg_state_migrate: p-adaptive runs are not combined with mesh refinement");
  // (matched on the device: the source ids sorted once, a binary search per destination row)
  return qdg::dev_state_migrate(from, from_gid, to, to_gid, nmoved);
  QDG_CATCH
}

static int rows_to_device(qdg_mesh* mesh, size_t n, const size_t* rows, DevBuf<int>& drow, const char* who)
{
  std::vector<int> d2h(mesh->ne), h2d(mesh->ne), dr(n);
  HIPCHK(hipMemcpy(d2h.data(), mesh->d2h.p, mesh->ne * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t d = 0; d < mesh->ne; ++d) h2d[d2h[d]] = (int)d;
  for (size_t j = 0; j < n; ++j) {
    if (rows[j] >= mesh->ne) return fail(std::string(who) + ": row id out of range");
    dr[j] = h2d[rows[j]];
  }
  HIPCHK(drow.upload(dr, mesh->ctx->stream));
  return 0;
}

extern "C" int qdg_state_rows_get(qdg_mesh* mesh, size_t n, const size_t* rows, void* packed_dev)
{
  QDG_TRY
  MESH_ENTER("qdg_state_rows_get");
  if (n == 0) return 0;
  if (!rows || !packed_dev) return fail("qdg_state_rows_get: null argument");
  DevBuf<int> drow;
  if (int rc = rows_to_device(mesh, n, rows, drow, "qdg_state_rows_get")) return rc;
  launch_rows_gather(n, mesh->nprop, drow.p, mesh->Ucur, static_cast<double*>(packed_dev), s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));        // drow is released on return
  return 0;
  QDG_CATCH
}

extern "C" int qdg_state_rows_put(qdg_mesh* mesh, size_t n, const size_t* rows, const void* packed_dev)
{
  QDG_TRY
  MESH_ENTER("qdg_state_rows_put");
  if (n == 0) return 0;
  if (!rows || !packed_dev) return fail("qdg_state_rows_put: null argument");
  DevBuf<int> drow;
  if (int rc = rows_to_device(mesh, n, rows, drow, "qdg_state_rows_put")) return rc;
  launch_rows_scatter(n, mesh->nprop, drow.p, static_cast<const double*>(packed_dev), mesh->Ucur, s);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(s));
  mesh->Unp = nullptr; mesh->Upending = nullptr;
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------- halo

// doubles per slab row: the row of U, plus the tet's ndof with p-adaptive DG
static size_t slab_w(const qdg_mesh* mesh) { return (size_t)mesh->nprop + (mesh->dm.ndofel ? 1 : 0); }

extern "C" int qdg_halo_setup(qdg_mesh* mesh, size_t nnbr, const int32_t* nbr_rank,
                              const size_t* send_off, const size_t* send_elem,
                              const size_t* recv_off)
{
  QDG_TRY
  MESH_ENTER("qdg_halo_setup");
  if (nnbr == 0) {
    // no neighbours: clear the halo state, touch none of the (possibly null) arrays
    if (mesh->ne != mesh->nie)
      return fail("qdg_halo_setup: the chunk has ghost rows but no neighbour was given");
    mesh->nnbr = mesh->nsend = mesh->nrecv = 0;
    mesh->nbr_rank.clear(); mesh->send_off.assign(1, 0); mesh->recv_off.assign(1, 0);
    return 0;
  }
  if (!nbr_rank || !send_off || !send_elem || !recv_off) return fail("qdg_halo_setup: null argument");
  if (send_off[0] != 0 || recv_off[0] != 0) return fail("qdg_halo_setup: offsets must start at 0");
  for (size_t i = 0; i < nnbr; ++i)
    if (send_off[i + 1] < send_off[i] || recv_off[i + 1] < recv_off[i])
      return fail("qdg_halo_setup: offsets must be non-decreasing");
  if (recv_off[nnbr] != mesh->ne - mesh->nie)
    return fail("qdg_halo_setup: receive counts must add up to the number of ghost rows");
  if (send_off[nnbr] > (size_t)INT32_MAX / 64) return fail("qdg_halo_setup: send list too long");
  mesh->nnbr = nnbr;
  mesh->nghost1 = 0; mesh->dm.nlim = (int)mesh->nie;      // a new plan: one ghost layer until qdg_halo_set_depth
  mesh->nbr_rank.assign(nbr_rank, nbr_rank + nnbr);
  mesh->send_off.assign(send_off, send_off + nnbr + 1);
  mesh->recv_off.assign(recv_off, recv_off + nnbr + 1);
  mesh->nsend = send_off[nnbr];
  mesh->nrecv = recv_off[nnbr];
  qdg::keep_set_plan(mesh, nnbr, nbr_rank, recv_off);      // (a handle that keeps its connectivity: for its re-mesh)
  // host tet id -> device row of the send list
  std::vector<int> d2h(mesh->ne), h2d(mesh->ne);
  HIPCHK(hipMemcpy(d2h.data(), mesh->d2h.p, mesh->ne * sizeof(int), hipMemcpyDeviceToHost));
  for (size_t d = 0; d < mesh->ne; ++d) h2d[d2h[d]] = (int)d;
  std::vector<int> se(mesh->nsend);
  for (size_t i = 0; i < mesh->nsend; ++i) {
    if (send_elem[i] >= mesh->nie) return fail("qdg_halo_setup: send list entry is not an interior tet");
    se[i] = h2d[send_elem[i]];
  }
  HIPCHK(mesh->send_elem.upload(se, s));
  // inverse of the send list for the packs folded into the producing kernels (qdg_step_comm): the slab rows
  // of every halo-adjacent device row.  Only where the plan has the shape the device order was built for --
  // every send row among the trailing rows [ninner, nie), at most FOLD_SLOTS slab rows per tet -- and not for
  // p-adaptive runs (their slab rows carry the ndof column)
  (void)mesh->fold_slot.alloc(0);
  mesh->slab_ready_for = nullptr;
  if (!mesh->dm.ndofel && mesh->nsend > 0) {
    const size_t ninner = (size_t)mesh->dm.ninner, nh = mesh->nie - ninner;
    std::vector<int> slot(FOLD_SLOTS * std::max<size_t>(nh, 1), -1);
    bool ok = true;
    for (size_t j = 0; j < mesh->nsend && ok; ++j) {
      const size_t d = (size_t)se[j];
      if (d < ninner) { ok = false; break; }
      int* sl = &slot[FOLD_SLOTS * (d - ninner)];
      int q = 0;
      while (q < FOLD_SLOTS && sl[q] >= 0) ++q;
      if (q == FOLD_SLOTS) ok = false; else sl[q] = (int)j;
    }
    if (ok) HIPCHK(mesh->fold_slot.upload(slot, s));
  }
  HIPCHK(mesh->send_slab.alloc(std::max<size_t>(1, mesh->nsend * slab_w(mesh))));
  HIPCHK(mesh->recv_slab.alloc(std::max<size_t>(1, mesh->nrecv * slab_w(mesh))));
  mesh->send_ptr = mesh->send_slab.p;
  mesh->recv_ptr = mesh->recv_slab.p;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_halo_set_depth(qdg_mesh* mesh, size_t nghost1)
{
  QDG_TRY
  MESH_ENTER("qdg_halo_set_depth");
  if (nghost1 == 0) { mesh->nghost1 = 0; mesh->dm.nlim = (int)mesh->nie; return 0; }
  if (nghost1 > mesh->ne - mesh->nie) return fail("qdg_halo_set_depth: more layer-1 ghosts than ghost rows");
  if (!mesh->ghost_nbr)
    return fail("qdg_halo_set_depth: the mesh holds no face neighbours of its ghost rows (build it with "
                "qdg_mesh_from_chunk[_gid]; qdg_mesh_upload takes esuel of the owned tets only)");
  if (mesh->dm.ndofel) return fail("qdg_halo_set_depth: p-adaptive DG runs with one ghost layer");
  if (mesh->nnbr == 0) return fail("qdg_halo_set_depth: call qdg_halo_setup first");
  mesh->nghost1 = nghost1;
  mesh->dm.nlim = (int)(mesh->nie + nghost1);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_halo_info(qdg_mesh* mesh, size_t* nentry, size_t* nghost1, int32_t* packs_folded)
{
  QDG_TRY
  if (!mesh) return fail("qdg_halo_info: null mesh");
  if (nentry) *nentry = mesh->nnbr;
  if (nghost1) *nghost1 = mesh->nghost1;
  if (packs_folded) *packs_folded = mesh->fold_slot.p ? 1 : 0;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_halo_buffers(qdg_mesh* mesh, void** send_dev, void** recv_dev, size_t* row_bytes)
{
  QDG_TRY
  if (!mesh || !send_dev || !recv_dev || !row_bytes) return fail("qdg_halo_buffers: null argument");
  *send_dev = mesh->send_ptr; *recv_dev = mesh->recv_ptr;
  *row_bytes = slab_w(mesh) * sizeof(double);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_halo_pack(qdg_mesh* mesh)
{
  QDG_TRY
  MESH_ENTER("qdg_halo_pack");
  launch_halo_pack(mesh->Ucur, mesh->nprop, (int)mesh->stride, mesh->send_elem.p, (int)mesh->nsend,
                   mesh->send_ptr, s, mesh->dm.ndofel);
  HIPCHK(hipGetLastError());
  return 0;
  QDG_CATCH
}

extern "C" int qdg_halo_unpack(qdg_mesh* mesh)
{
  QDG_TRY
  MESH_ENTER("qdg_halo_unpack");
  launch_halo_unpack(mesh->recv_ptr, mesh->nprop, (int)mesh->stride, (int)mesh->nie,
                     (int)mesh->nrecv, mesh->Ucur, s, mesh->ndofel.p);
  HIPCHK(hipGetLastError());
  return 0;
  QDG_CATCH
}

// chunks that live on the same device under one context (several chares of one PE in the
// reference exchange through memory too): rows [src_row0, src_row0+nrows) of src's send slab ->
// rows [dst_row0, ...) of dst's receive slab, on the context's stream
extern "C" int qdg_halo_copy(qdg_mesh* dst, size_t dst_row0, qdg_mesh* src, size_t src_row0, size_t nrows)
{
  QDG_TRY
  if (!dst || !src) return fail("qdg_halo_copy: null mesh");
  if (dst->ctx != src->ctx) return fail("qdg_halo_copy: both chunks must belong to one context");
  if (slab_w(dst) != slab_w(src)) return fail("qdg_halo_copy: row length differs");
  if (src_row0 + nrows > src->nsend || dst_row0 + nrows > dst->nrecv) return fail("qdg_halo_copy: row range outside the slabs");
  if (nrows == 0) return 0;
  HIPCHK(hipSetDevice(dst->ctx->device));
  const size_t w = slab_w(dst);
  HIPCHK(hipMemcpyAsync(dst->recv_ptr + dst_row0 * w, src->send_ptr + src_row0 * w, nrows * w * sizeof(double),
                        hipMemcpyDeviceToDevice, dst->ctx->stream));
  dst->slab_ready_for = src->slab_ready_for = nullptr;    // a caller-driven exchange: qdg_step_comm packs afresh
  return 0;
  QDG_CATCH
}

extern "C" int qdg_halo_use_buffers(qdg_mesh* mesh, void* send_dev, void* recv_dev)
{
  QDG_TRY
  if (!mesh) return fail("qdg_halo_use_buffers: null mesh");
  if ((mesh->nsend && !send_dev) || (mesh->nrecv && !recv_dev))
    return fail("qdg_halo_use_buffers: null buffer");
  mesh->send_ptr = (double*)send_dev;
  mesh->recv_ptr = (double*)recv_dev;
  mesh->slab_ready_for = nullptr;      // another send slab: nothing packed in it yet
  return 0;
  QDG_CATCH
}

extern "C" int qdg_halo_sizes(qdg_mesh* mesh, size_t* nsend_rows, size_t* nrecv_rows)
{
  QDG_TRY
  if (!mesh || !nsend_rows || !nrecv_rows) return fail("qdg_halo_sizes: null argument");
  *nsend_rows = mesh->nsend; *nrecv_rows = mesh->nrecv;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_stage_dt_use_buffer(qdg_mesh* mesh, void* dt_dev)
{
  QDG_TRY
  if (!mesh || !dt_dev) return fail("qdg_stage_dt_use_buffer: null argument");
  mesh->dt_ptr = (double*)dt_dev;
  return 0;
  QDG_CATCH
}

// ---------------------------------------------------------------- RCCL transport

namespace {
// RCCL entry points, resolved at run time so that libqdg has no link-time
// dependency on a particular librccl (a process that already loaded one, e.g.
// through PyTorch, keeps using that copy)
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
  std::string error;
};

RcclApi* rccl_api()
{
  // resolved once; thread-safe (function-local static initialised by a lambda)
  static RcclApi* const instance = [] {
    static RcclApi api;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) {
      api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) { api.error = std::string("cannot load librccl: ") + dlerror(); return &api; }
#define QDG_RCCL_SYM(f)                                                        \
    api.f = reinterpret_cast<decltype(api.f)>(dlsym(api.handle, "nccl" #f));   \
    if (!api.f) { api.error = "librccl lacks nccl" #f; return &api; }
    QDG_RCCL_SYM(GetUniqueId) QDG_RCCL_SYM(CommInitRank) QDG_RCCL_SYM(CommDestroy)
    QDG_RCCL_SYM(GetErrorString) QDG_RCCL_SYM(GroupStart) QDG_RCCL_SYM(GroupEnd)
    QDG_RCCL_SYM(Send) QDG_RCCL_SYM(Recv) QDG_RCCL_SYM(AllReduce)
    QDG_RCCL_SYM(CommCount) QDG_RCCL_SYM(CommUserRank) QDG_RCCL_SYM(CommCuDevice)
#undef QDG_RCCL_SYM
    return &api;
  }();
  return instance;
}
}  // namespace

#define RCCLCHK(call)                                                             \
  do {                                                                            \
    ncclResult_t r_ = (call);                                                     \
    if (r_ != ncclSuccess)                                                        \
      return ::qdg::fail(std::string(#call) + ": " + rccl_api()->GetErrorString(r_)); \
  } while (0)

struct qdg_comm {
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0, device = 0;
  qdg_comm() = default;
  qdg_comm(const qdg_comm&) = delete;
  qdg_comm& operator=(const qdg_comm&) = delete;
  // releases whatever was created, on every path (a failed qdg_comm_create included)
  ~qdg_comm()
  {
    (void)hipSetDevice(device);
    if (comm) (void)rccl_api()->CommDestroy(comm);
  }
};

extern "C" int qdg_comm_unique_id(void* id128)
{
  QDG_TRY
  if (!id128) return fail("qdg_comm_unique_id: null argument");
  RcclApi* a = rccl_api();
  if (!a->error.empty()) return fail("qdg_comm_unique_id: " + a->error);
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the ABI of qdg.h");
  ncclUniqueId id;
  RCCLCHK(a->GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof id);
  return 0;
  QDG_CATCH
}

extern "C" int qdg_comm_create(qdg_ctx* ctx, int nranks, int rank, const void* id128, qdg_comm** out)
{
  QDG_TRY
  if (!ctx || !id128 || !out) return fail("qdg_comm_create: null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail("qdg_comm_create: bad rank / nranks");
  RcclApi* a = rccl_api();
  if (!a->error.empty()) return fail("qdg_comm_create: " + a->error);
  HIPCHK(hipSetDevice(ctx->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  std::unique_ptr<qdg_comm> c(new qdg_comm);
  c->nranks = nranks; c->rank = rank; c->device = ctx->device;
  // RCCL allocates from the driver and never sees this library's cache of freed blocks: hand it back first
  (void)qdg::DevicePool::get().trim();
  RCCLCHK(a->CommInitRank(&c->comm, nranks, id, rank));
  *out = c.release();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_comm_info(qdg_comm* comm, int* nranks, int* rank, int* device)
{
  QDG_TRY
  if (!comm || !comm->comm) return fail("qdg_comm_info: null communicator");
  RcclApi* a = rccl_api();
  int n = 0, r = 0, d = 0;
  RCCLCHK(a->CommCount(comm->comm, &n));
  RCCLCHK(a->CommUserRank(comm->comm, &r));
  RCCLCHK(a->CommCuDevice(comm->comm, &d));
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (device) *device = d;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_comm_destroy(qdg_comm* comm)
{
  QDG_TRY
  if (!comm) return 0;
  ncclResult_t r = ncclSuccess;
  if (comm->comm) {
    (void)hipSetDevice(comm->device);
    r = rccl_api()->CommDestroy(comm->comm);
    comm->comm = nullptr;
  }
  delete comm;                       // stream and events go with it, whatever CommDestroy said
  if (r != ncclSuccess) return fail(std::string("qdg_comm_destroy: ") + rccl_api()->GetErrorString(r));
  return 0;
  QDG_CATCH
}

// pack, grouped send / receive into the ghost rows -- all enqueued on stream `s`
static int exchange_on_impl(qdg_mesh* mesh, qdg_comm* comm, hipStream_t s, bool packed);
static int exchange_upd_impl(qdg_mesh* mesh, qdg_comm* comm, hipStream_t s, double* out);

// (with qdg_profile_enable: an event pair around the whole exchange -- pack kernel, RCCL kernel and the gaps
// in front of them, which is what the exchange costs the step)
static int exchange_on(qdg_mesh* mesh, qdg_comm* comm, hipStream_t s, bool packed = false)
{
  if (mesh->nnbr == 0) return 0;
  std::pair<hipEvent_t, hipEvent_t>* ev;
  if (int rc = prof_begin(mesh, &ev, false, 1)) return rc;
  if (int rc = exchange_on_impl(mesh, comm, s, packed)) return rc;
  if (ev) HIPCHK(hipEventRecord(ev->second, s));
  return 0;
}

static int exchange_upd(qdg_mesh* mesh, qdg_comm* comm, hipStream_t s, double* out)
{
  if (mesh->nnbr == 0) return 0;
  std::pair<hipEvent_t, hipEvent_t>* ev;
  if (int rc = prof_begin(mesh, &ev, false, 1)) return rc;
  if (int rc = exchange_upd_impl(mesh, comm, s, out)) return rc;
  if (ev) HIPCHK(hipEventRecord(ev->second, s));
  return 0;
}

static int exchange_on_impl(qdg_mesh* mesh, qdg_comm* comm, hipStream_t s, bool packed)
{
  if (mesh->nnbr == 0) return 0;
  if (!comm) return fail("qdg_halo_exchange: null communicator");
  for (size_t i = 0; i < mesh->nnbr; ++i)
    if (mesh->nbr_rank[i] < 0 || mesh->nbr_rank[i] >= comm->nranks)
      return fail("qdg_halo_exchange: neighbour rank outside the communicator");
  RcclApi* a = rccl_api();
  const size_t np = (size_t)mesh->nprop, w = slab_w(mesh);
  const bool direct = w == np;       // ghost rows are contiguous per neighbour: received in place
  if (!packed) {                     // (packed: the kernel that produced the state has filled the slab)
    launch_halo_pack(mesh->Ucur, mesh->nprop, (int)mesh->stride, mesh->send_elem.p, (int)mesh->nsend,
                     mesh->send_ptr, s, mesh->dm.ndofel);
    HIPCHK(hipGetLastError());
  }
  RCCLCHK(a->GroupStart());
  // an error inside the group must not leave it open: remember the first one, always
  // close the group, then report
  ncclResult_t first = ncclSuccess;
  for (size_t i = 0; i < mesh->nnbr && first == ncclSuccess; ++i) {
    const size_t ns = (mesh->send_off[i + 1] - mesh->send_off[i]) * w;
    const size_t nr = (mesh->recv_off[i + 1] - mesh->recv_off[i]) * w;
    double* dst = direct ? mesh->Ucur + (mesh->nie + mesh->recv_off[i]) * np
                         : mesh->recv_ptr + mesh->recv_off[i] * w;
    if (ns) first = a->Send(mesh->send_ptr + mesh->send_off[i] * w, ns, ncclDouble, mesh->nbr_rank[i], comm->comm, s);
    if (nr && first == ncclSuccess) first = a->Recv(dst, nr, ncclDouble, mesh->nbr_rank[i], comm->comm, s);
  }
  const ncclResult_t endr = a->GroupEnd();
  if (first != ncclSuccess) return fail(std::string("qdg_halo_exchange: ncclSend/ncclRecv: ") + a->GetErrorString(first));
  if (endr != ncclSuccess) return fail(std::string("qdg_halo_exchange: ncclGroupEnd: ") + a->GetErrorString(endr));
  if (!direct) {                     // p-adaptive DG: rows carry the ndof column
    launch_halo_unpack(mesh->recv_ptr, mesh->nprop, (int)mesh->stride, (int)mesh->nie,
                       (int)mesh->nrecv, mesh->Ucur, s, mesh->ndofel.p);
    HIPCHK(hipGetLastError());
  }
  mesh->carry_src = nullptr;         // the ghost rows of the current state have been received
  return 0;
}

// the comsol exchange of stage 1 when the stage-0 update is fused into the limiter:
// the send rows U1 = U0 + dt R / L are formed by the pack kernel, the neighbours' rows
// land in the ghost rows of `out`, the buffer the fused kernel is about to fill
static int exchange_upd_impl(qdg_mesh* mesh, qdg_comm* comm, hipStream_t s, double* out)
{
  if (mesh->nnbr == 0) return 0;
  if (!comm) return fail("qdg_step_comm: null communicator");
  RcclApi* a = rccl_api();
  const size_t np = (size_t)mesh->nprop;
  launch_halo_pack_upd(mesh->Ucur, mesh->R.p, mesh->dt_ptr, mesh->vol.p, mesh->send_elem.p,
                       (int)mesh->nsend, mesh->send_ptr, s);
  HIPCHK(hipGetLastError());
  RCCLCHK(a->GroupStart());
  ncclResult_t first = ncclSuccess;
  for (size_t i = 0; i < mesh->nnbr && first == ncclSuccess; ++i) {
    const size_t ns = (mesh->send_off[i + 1] - mesh->send_off[i]) * np;
    const size_t nr = (mesh->recv_off[i + 1] - mesh->recv_off[i]) * np;
    if (ns) first = a->Send(mesh->send_ptr + mesh->send_off[i] * np, ns, ncclDouble, mesh->nbr_rank[i], comm->comm, s);
    if (nr && first == ncclSuccess)
      first = a->Recv(out + (mesh->nie + mesh->recv_off[i]) * np, nr, ncclDouble, mesh->nbr_rank[i], comm->comm, s);
  }
  const ncclResult_t endr = a->GroupEnd();
  if (first != ncclSuccess) return fail(std::string("qdg_step_comm: ncclSend/ncclRecv: ") + a->GetErrorString(first));
  if (endr != ncclSuccess) return fail(std::string("qdg_step_comm: ncclGroupEnd: ") + a->GetErrorString(endr));
  return 0;
}

extern "C" int qdg_halo_exchange(qdg_mesh* mesh, qdg_comm* comm)
{
  QDG_TRY
  MESH_ENTER("qdg_halo_exchange");
  return exchange_on(mesh, comm, s, false);
  QDG_CATCH
}

extern "C" int qdg_stage_dt_allreduce(qdg_mesh* mesh, qdg_comm* comm)
{
  QDG_TRY
  MESH_ENTER("qdg_stage_dt_allreduce");
  if (!comm) return fail("qdg_stage_dt_allreduce: null communicator");
  std::pair<hipEvent_t, hipEvent_t>* ev;
  if (int rc = prof_begin(mesh, &ev, false, 2)) return rc;
  RCCLCHK(rccl_api()->AllReduce(mesh->dt_ptr, mesh->dt_ptr, 1, ncclDouble, ncclMin, comm->comm, s));
  if (ev) HIPCHK(hipEventRecord(ev->second, s));
  return 0;
  QDG_CATCH
}

static int step_comm_stages(qdg_mesh* mesh, qdg_comm* comm, double t, double tleft, bool slab_ready);

// ---- qdg_step_comm as a hipGraph (context option graph_step) --------------------------------------
// The launch sequence of a step -- ~9 compute kernels, the RCCL send / receive and all-reduce kernels, the
// ghost-row copy -- depends on nothing but WHICH of the three state buffers is current at entry (the
// rotation has period two) and on the arguments t and tleft.  It is captured once per such entry state and
// replayed with ONE hipGraphLaunch: the host enqueues one command per step instead of ~25, and the runtime
// knows the whole dependency chain ahead of time.  Conditions: the RHS must not read t (Sod, Sedov and
// rotated Sod: no source term, no analytic boundary state -- else every step has its own t and nothing can
// be replayed), uniform order (the p-adaptive step has a host-visible phase), no profiling events.  RCCL sets
// its peer connections up lazily inside the first exchanges, so the first two steps of a mesh run plain.
// A capture that is refused (by HIP or by RCCL) leaves the plain path in charge; qdg_step_graph_status says why.
static bool step_reads_t(const qdg_ctx* ctx)
{
  if (ctx->cfg.pde != QDG_PDE_COMPFLOW) return true;
  return !(ctx->cfg.problem == QDG_PROBLEM_SOD_SHOCKTUBE || ctx->cfg.problem == QDG_PROBLEM_SEDOV_BLASTWAVE ||
           ctx->cfg.problem == QDG_PROBLEM_ROTATED_SOD_SHOCKTUBE);
}

struct StepState {             // what step_comm_stages changes in the handle
  double *Ucur, *Unp, *Upending, *carry_src; const double* slab_ready_for; bool skip_ghost_carry;
  explicit StepState(const qdg_mesh* m) : Ucur(m->Ucur), Unp(m->Unp), Upending(m->Upending), carry_src(m->carry_src),
                                          slab_ready_for(m->slab_ready_for), skip_ghost_carry(m->skip_ghost_carry) {}
  void restore(qdg_mesh* m) const
  {
    m->Ucur = Ucur; m->Unp = Unp; m->Upending = Upending; m->carry_src = carry_src;
    m->slab_ready_for = slab_ready_for; m->skip_ghost_carry = skip_ghost_carry;
  }
};

// 1: the step was enqueued as a graph launch; 0: not applicable / refused, take the plain path; < 0: error
static int step_comm_graph(qdg_mesh* mesh, qdg_comm* comm, double t, double tleft, bool slab_ready)
{
  qdg_ctx* ctx = mesh->ctx;
  hipStream_t s = ctx->stream;
  if (!ctx->opt.graph_step || mesh->graph_state < 0 || mesh->prof || mesh->dm.ndofel || step_reads_t(ctx)) return 0;
  if (mesh->Unp || mesh->Upending) return 0;                 // a stage in flight: not the entry state of a step
  if (mesh->graph_warm < 2) { ++mesh->graph_warm; return 0; }
  for (const qdg_mesh::StepGraph& g : mesh->step_graphs)
    if (g.ucur_in == mesh->Ucur && g.tleft == tleft && g.slab_ready_in == slab_ready) {
      if (hipGraphLaunch(g.exec, s) != hipSuccess) return -1;
      mesh->Ucur = g.ucur_out; mesh->Unp = nullptr; mesh->Upending = nullptr; mesh->carry_src = g.carry_out;
      mesh->slab_ready_for = g.slab_ready_out;
      ++mesh->graph_replays;
      return 1;
    }
  if (mesh->step_graphs.size() >= 8) return 0;               // (tleft changes every step: the caller's last steps)
  const StepState before(mesh);
  auto refuse = [&](const std::string& why) {
    mesh->graph_state = -1;
    mesh->graph_error = why;
    before.restore(mesh);
    return 0;
  };
  hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) return refuse(std::string("hipStreamBeginCapture: ") + hipGetErrorString(e));
  const int rc = step_comm_stages(mesh, comm, t, tleft, slab_ready);
  const std::string rc_msg = rc ? qdg_last_error() : "";
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(s, &graph);
  if (rc || e != hipSuccess || !graph) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    return refuse(rc ? "capture of the step: " + rc_msg : std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess || !exec) { (void)hipGetLastError(); return refuse(std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
  mesh->step_graphs.push_back({ before.Ucur, t, tleft, slab_ready, exec, mesh->Ucur, mesh->slab_ready_for, mesh->carry_src });
  mesh->graph_state = 1;
  // (nothing has run yet: the capture only recorded; the handle's state is already the state after the step)
  if (hipGraphLaunch(exec, s) != hipSuccess) { before.restore(mesh); return -1; }
  ++mesh->graph_replays;
  return 1;
}

extern "C" int qdg_step_comm(qdg_mesh* mesh, qdg_comm* comm, double t, double tleft, double* dt_taken)
{
  QDG_TRY
  // (read before MESH_ENTER clears it: the slab still holds this state's send rows if the previous call on
  // this mesh was a qdg_step_comm whose last kernel packed them)
  const bool slab_ready = mesh && mesh->slab_ready_for && mesh->slab_ready_for == mesh->Ucur;
  // (a ghost-row carry left pending by the previous qdg_step_comm is dropped: this step starts by receiving every
  // ghost row -- except the p-adaptive step, whose first kernel reads them)
  const bool drop_carry = mesh && !mesh->dm.ndofel && mesh->nnbr > 0 && mesh->nrecv == mesh->ne - mesh->nie && comm;
  if (drop_carry) mesh->carry_pending = nullptr;
  MESH_ENTER("qdg_step_comm");
  if (!comm) return fail("qdg_step_comm: null communicator");
  // the skipped ghost-row carries below rely on every ghost row being received by the next exchange
  if (mesh->nnbr > 0 && mesh->nrecv != mesh->ne - mesh->nie)
    return fail("qdg_step_comm: the halo plan does not cover every ghost row");
  const int g = step_comm_graph(mesh, comm, t, tleft, slab_ready);
  if (g < 0) return fail("qdg_step_comm: hipGraphLaunch failed");
  const int rc = g == 1 ? 0 : step_comm_stages(mesh, comm, t, tleft, slab_ready);
  const double* ready_after = mesh->slab_ready_for;
  if (rc && mesh->carry_src && mesh->carry_src != mesh->Ucur) {
    // an exchange failed after an update whose ghost-row carry was skipped: do the carry now, so that
    // the state the caller is left with has the last received ghost rows, not a buffer's older content
    const std::string msg = qdg_last_error();
    (void)hipMemcpyAsync(mesh->Ucur + mesh->nie * (size_t)mesh->nprop, mesh->carry_src + mesh->nie * (size_t)mesh->nprop,
                         (mesh->ne - mesh->nie) * (size_t)mesh->nprop * sizeof(double), hipMemcpyDeviceToDevice, s);
    mesh->carry_src = nullptr;
    return fail(msg);
  }
  // the last update's ghost-row carry is left to whoever looks at the ghost rows next (flush_ghost_carry)
  if (!rc && mesh->carry_src && mesh->carry_src != mesh->Ucur) mesh->carry_pending = mesh->carry_src;
  mesh->carry_src = nullptr;
  if (rc) return rc;
  if (dt_taken) {
    const int rc2 = qdg_stage_dt_get(mesh, dt_taken);      // (its MESH_ENTER clears the mark)
    mesh->slab_ready_for = ready_after;
    return rc2;
  }
  return 0;
  QDG_CATCH
}

static int step_comm_stages(qdg_mesh* mesh, qdg_comm* comm, double t, double tleft, bool slab_ready)
{
  qdg_ctx* ctx = mesh->ctx;
  hipStream_t s = ctx->stream;
  const bool limited = ctx->cfg.limiter != QDG_LIMITER_NONE && mesh->ndof > 1;
  const bool fuse = can_fuse_update_limit(mesh);
  // Packs folded into the kernels that produce the rows (halo_fold_row): the Superbee kernels write the
  // limited send rows to the slab themselves, the DG-P1 tile kernel with the fused RK update the rows of the
  // new state.  Enabled for THIS call's launches only (DevMesh goes to the kernels by value).
  const bool can_fold = mesh->fold_slot.p && mesh->nnbr > 0;
  // (only the CompFlow Superbee kernels fold: tr::k_superbee of dg::Transport does not write the slab)
  // Two ghost layers (qdg_halo_set_depth): the limiter also covers the layer-1 ghosts, whose inputs layer 2
  // completes, so the exchange of the limited solution (comlim) is not needed -- 3 exchanges per step, not 6
  const bool deep = mesh->nghost1 > 0;
  const bool lim_folds = can_fold && ctx->cfg.limiter == QDG_LIMITER_SUPERBEEP1 && mesh->ndof > 1 &&
                         mesh->dm.pde == QDG_PDE_COMPFLOW && !deep;
  const bool rhs_folds = can_fold && use_p1_fast(mesh) && use_tile(mesh) && !mesh->dm.ndofel;
  struct FoldScope {
    qdg_mesh* m;
    FoldScope(qdg_mesh* mm, bool on) : m(mm) { if (on) { m->dm.fold_slot = m->fold_slot.p; m->dm.fold_slab = m->send_ptr; } }
    ~FoldScope() { m->dm.fold_slot = nullptr; m->dm.fold_slab = nullptr; }
  } fold_scope(mesh, can_fold);
  bool packed = slab_ready && rhs_folds;       // the previous step's last kernel packed this state's send rows
  for (int stage = 0; stage < 3; ++stage) {
    const bool pdg0 = stage == 0 && mesh->dm.ndofel;
    const bool fused1 = fuse && stage == 1;        // comsol + limiter of stage 1 ran with the update
    if (pdg0) if (int rc = qdg_stage_pdg_eval(mesh)) return rc;          // DG::next: eval_ndof
    if (!fused1) if (int rc = exchange_on(mesh, comm, s, packed && !pdg0)) return rc;    // DG::next -> comsol
    packed = false;
    if (pdg0) if (int rc = qdg_stage_pdg_propagate(mesh)) return rc;     // DG::lim: propagate_ndof
    if (!fused1) if (int rc = qdg_stage_limit(mesh)) return rc;          // DG::lim
    if ((limited && !deep) || pdg0) if (int rc = exchange_on(mesh, comm, s, lim_folds && !pdg0)) return rc;   // -> comlim
    if (int rc = qdg_stage_rhs_dt(mesh, stage, t, tleft)) return rc;     // DG::dt, DG::solve
    if (stage == 0) if (int rc = qdg_stage_dt_allreduce(mesh, comm)) return rc;
    if (fuse && stage == 0) {
      // update of stage 0 + comsol + limiter of stage 1 in one pass over the state
      double* out = free_buf(mesh, mesh->Ucur, mesh->Unp);
      if (int rc = exchange_upd(mesh, comm, s, out)) return rc;
      // (two ghost layers: the same launch also limits the layer-1 ghosts, whose unlimited U1 has just arrived)
      if (int rc = stage0_update_and_limit(mesh)) return rc;
    } else {
      // stages 0, 1: the next stage starts by receiving the ghost rows of the new state; stage 2: so does the
      // next step -- the carry stays pending for any other reader (qdg_step_comm, flush_ghost_carry)
      mesh->skip_ghost_carry = mesh->nnbr > 0 && (stage < 2 || !mesh->dm.ndofel);
      const bool by_kernel = mesh->Upending != nullptr;      // the fused RHS + RK kernel wrote the new state
      const int rc = qdg_stage_update(mesh, stage);
      mesh->skip_ghost_carry = false;
      if (rc) return rc;
      packed = by_kernel && rhs_folds;                       // ... and with it the next comsol's send rows
    }
  }
  // (qdg_step_comm hands this to the next call if nothing else touches the mesh in between)
  mesh->slab_ready_for = packed ? mesh->Ucur : nullptr;
  return 0;
}

// ---------------------------------------------------------------- measurement

extern "C" int qdg_profile_enable(qdg_mesh* mesh, int on)
{
  QDG_TRY
  if (!mesh) return fail("qdg_profile_enable: null mesh");
  mesh->prof = on != 0;
  mesh->ev_used = 0;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_profile_read(qdg_mesh* mesh, size_t* nlaunch, double* total_ms)
{
  QDG_TRY
  MESH_ENTER_NOFLUSH("qdg_profile_read");
  if (!nlaunch || !total_ms) return fail("qdg_profile_read: null argument");
  HIPCHK(hipStreamSynchronize(s));
  double tot = 0.0;
  size_t nl = 0;
  for (size_t i = 0; i < mesh->ev_used; ++i) {
    if (mesh->ev_kind[i] != 0) continue;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, mesh->ev[i].first, mesh->ev[i].second));
    tot += ms;
    if (!mesh->ev_cont[i]) ++nl;
  }
  *nlaunch = nl;
  *total_ms = tot;
  mesh->ev_used = 0;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_profile_read_all(qdg_mesh* mesh, size_t count[3], double total_ms[3])
{
  QDG_TRY
  MESH_ENTER_NOFLUSH("qdg_profile_read_all");
  if (!count || !total_ms) return fail("qdg_profile_read_all: null argument");
  HIPCHK(hipStreamSynchronize(s));
  for (int k = 0; k < 3; ++k) { count[k] = 0; total_ms[k] = 0.0; }
  for (size_t i = 0; i < mesh->ev_used; ++i) {
    const int k = mesh->ev_kind[i];
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, mesh->ev[i].first, mesh->ev[i].second));
    total_ms[k] += ms;
    if (!mesh->ev_cont[i]) ++count[k];
  }
  mesh->ev_used = 0;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_step_graph_status(qdg_mesh* mesh, int32_t* state, int32_t* ngraphs, int64_t* nreplays,
                                     char* error, size_t error_len)
{
  QDG_TRY
  if (!mesh) return fail("qdg_step_graph_status: null mesh");
  if (state) *state = mesh->graph_state;
  if (ngraphs) *ngraphs = (int32_t)mesh->step_graphs.size();
  if (nreplays) *nreplays = mesh->graph_replays;
  if (error && error_len) {
    std::strncpy(error, mesh->graph_error.c_str(), error_len - 1);
    error[error_len - 1] = 0;
  }
  return 0;
  QDG_CATCH
}

extern "C" int qdg_mesh_layout_stats(qdg_mesh* mesh, size_t counts[4])
{
  QDG_TRY
  MESH_ENTER_NOFLUSH("qdg_mesh_layout_stats");
  if (!counts) return fail("qdg_mesh_layout_stats: null argument");
  counts[0] = counts[1] = counts[2] = 0;
  counts[3] = (size_t)mesh->dm.ntile;
  std::vector<int> ta(mesh->task_a.n);
  if (!ta.empty()) {
    HIPCHK(hipMemcpyAsync(ta.data(), mesh->task_a.p, ta.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  for (int a : ta) if (a >= 0) ++counts[TASK_KIND(a)];
  return 0;
  QDG_CATCH
}

extern "C" int qdg_rhs_algorithmic_bytes(qdg_mesh* mesh, double* bytes)
{
  QDG_TRY
  if (!mesh || !bytes) return fail("qdg_rhs_algorithmic_bytes: null argument");
  *bytes = (double)mesh->nie * (16.0 * mesh->nprop + 32.0) + 24.0 * (double)mesh->nnode_used;
  return 0;
  QDG_CATCH
}

