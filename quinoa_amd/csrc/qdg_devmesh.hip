// qdg_devmesh.hip -- mesh-derived data of one chunk generated ON THE DEVICE
// (first step): what inciter::FaceData's constructor and the
// geometry generators produce (src/Inciter/FaceData.cpp:19-41,
// src/Mesh/DerivedData.cpp:937-1491), with the same content and the same order
// as the host mirror in qdg_meshdata.cpp -- the integer arrays bit for bit.
//
//   faces of all tets -> keys (sorted node triple) -> 3 stable radix sorts
//   (rocPRIM, least significant node first) -> equal neighbours in the sorted
//   order are the two sides of an interior face -> esuel;  interior faces are
//   numbered by an exclusive scan over (element, local face) in the reference's
//   order (kept when element < neighbour);  boundary faces find their element by
//   binary search in the sorted keys;  geometry is one thread per face / tet.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/qdg.h"
#include "qdg_host.hpp"

namespace qdg {
// defined in qdg_api.cpp
int ctx_device(const qdg_ctx* ctx);
hipStream_t ctx_stream(const qdg_ctx* ctx);
}  // namespace qdg

using namespace qdg;

namespace {

#define DHIP(call)                                                                \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess)                                                         \
      return ::qdg::fail(std::string(#call) + ": " + hipGetErrorString(e_));      \
  } while (0)

template <class T> struct Buf {
  T* p = nullptr;
  ~Buf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)); }
};

__constant__ int c_lpofa[4][3] = { { 1, 2, 3 }, { 2, 0, 3 }, { 3, 0, 1 }, { 0, 2, 1 } };

__global__ void k_face_keys(const uint64_t* __restrict__ inpoel, size_t n4, uint32_t* __restrict__ a,
                            uint32_t* __restrict__ b, uint32_t* __restrict__ c,
                            uint32_t* __restrict__ perm)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const size_t e = i >> 2; const int f = (int)(i & 3);
  uint32_t k0 = (uint32_t)inpoel[4 * e + c_lpofa[f][0]], k1 = (uint32_t)inpoel[4 * e + c_lpofa[f][1]],
           k2 = (uint32_t)inpoel[4 * e + c_lpofa[f][2]], t;
  if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
  if (k1 > k2) { t = k1; k1 = k2; k2 = t; }
  if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
  a[i] = k0; b[i] = k1; c[i] = k2; perm[i] = (uint32_t)i;
}

__global__ void k_gather(const uint32_t* __restrict__ src, const uint32_t* __restrict__ perm, size_t n,
                         uint32_t* __restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[perm[i]];
}

// sa, sb, sc: keys in sorted order; perm[i] = 4*element + local face of sorted entry i
__global__ void k_match(const uint32_t* __restrict__ sa, const uint32_t* __restrict__ sb,
                        const uint32_t* __restrict__ sc, const uint32_t* __restrict__ perm, size_t n,
                        int* __restrict__ esuel, int* __restrict__ err)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool eq_next = i + 1 < n && sa[i] == sa[i + 1] && sb[i] == sb[i + 1] && sc[i] == sc[i + 1];
  const bool eq_prev = i > 0 && sa[i] == sa[i - 1] && sb[i] == sb[i - 1] && sc[i] == sc[i - 1];
  if (eq_next && eq_prev) *err = 1;                 // a face shared by more than two tets
  int v = -1;
  if (eq_next) v = (int)(perm[i + 1] >> 2);
  else if (eq_prev) v = (int)(perm[i - 1] >> 2);
  esuel[perm[i]] = v;
}

__global__ void k_flag(const int* __restrict__ esuel, size_t n4, int* __restrict__ flag)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int j = esuel[i];
  flag[i] = (j != -1 && (int)(i >> 2) < j) ? 1 : 0;
}

__global__ void k_interior_faces(const uint64_t* __restrict__ inpoel, const int* __restrict__ esuel,
                                 const int* __restrict__ flag, const int* __restrict__ pos, size_t n4,
                                 size_t nbfac, uint64_t* __restrict__ inpofa, int* __restrict__ esuf)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4 || !flag[i]) return;
  const size_t e = i >> 2, fid = nbfac + (size_t)pos[i]; const int f = (int)(i & 3);
  inpofa[3 * fid] = inpoel[4 * e + c_lpofa[f][0]];
  inpofa[3 * fid + 1] = inpoel[4 * e + c_lpofa[f][1]];
  inpofa[3 * fid + 2] = inpoel[4 * e + c_lpofa[f][2]];
  esuf[2 * fid] = (int)e;
  esuf[2 * fid + 1] = esuel[i];
}

// boundary faces: inpofa = triinpoel; host element by binary search in the sorted keys
__global__ void k_boundary_faces(const uint64_t* __restrict__ tri, size_t nbfac,
                                 const uint32_t* __restrict__ sa, const uint32_t* __restrict__ sb,
                                 const uint32_t* __restrict__ sc, const uint32_t* __restrict__ perm,
                                 size_t n, uint64_t* __restrict__ inpofa, int* __restrict__ esuf,
                                 uint64_t* __restrict__ belem, int* __restrict__ err)
{
  const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nbfac) return;
  uint32_t k0 = (uint32_t)tri[3 * f], k1 = (uint32_t)tri[3 * f + 1], k2 = (uint32_t)tri[3 * f + 2], t;
  inpofa[3 * f] = tri[3 * f]; inpofa[3 * f + 1] = tri[3 * f + 1]; inpofa[3 * f + 2] = tri[3 * f + 2];
  if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
  if (k1 > k2) { t = k1; k1 = k2; k2 = t; }
  if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
  size_t lo = 0, hi = n;                               // first entry >= (k0, k1, k2)
  while (lo < hi) {
    const size_t mid = (lo + hi) >> 1;
    const bool less = sa[mid] < k0 || (sa[mid] == k0 && (sb[mid] < k1 || (sb[mid] == k1 && sc[mid] < k2)));
    if (less) lo = mid + 1; else hi = mid;
  }
  if (lo >= n || sa[lo] != k0 || sb[lo] != k1 || sc[lo] != k2) { *err = 2; return; }
  const uint64_t e = perm[lo] >> 2;
  belem[f] = e;
  esuf[2 * f] = (int)e;
  esuf[2 * f + 1] = -1;
}

// src/Mesh/DerivedData.cpp:1292-1434: area by Heron's formula, unit normal, centroid
__global__ void k_geoface(const uint64_t* __restrict__ inpofa, size_t nfac, const double* __restrict__ x,
                          const double* __restrict__ y, const double* __restrict__ z,
                          double* __restrict__ geoFace)
{
  const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nfac) return;
  const uint64_t a = inpofa[3 * f], b = inpofa[3 * f + 1], c = inpofa[3 * f + 2];
  const double X[3] = { x[a], x[b], x[c] }, Y[3] = { y[a], y[b], y[c] }, Z[3] = { z[a], z[b], z[c] };
  const double sa = sqrt((X[1]-X[0])*(X[1]-X[0]) + (Y[1]-Y[0])*(Y[1]-Y[0]) + (Z[1]-Z[0])*(Z[1]-Z[0]));
  const double sb = sqrt((X[2]-X[1])*(X[2]-X[1]) + (Y[2]-Y[1])*(Y[2]-Y[1]) + (Z[2]-Z[1])*(Z[2]-Z[1]));
  const double sc = sqrt((X[0]-X[2])*(X[0]-X[2]) + (Y[0]-Y[2])*(Y[0]-Y[2]) + (Z[0]-Z[2])*(Z[0]-Z[2]));
  const double sp = 0.5 * (sa + sb + sc);
  const double ax = X[1]-X[0], ay = Y[1]-Y[0], az = Z[1]-Z[0];
  const double bx = X[2]-X[0], by = Y[2]-Y[0], bz = Z[2]-Z[0];
  const double nx = ay * bz - az * by, ny = -(ax * bz - az * bx), nz = ax * by - ay * bx;
  const double fa = sqrt(nx * nx + ny * ny + nz * nz);
  double* g = geoFace + 7 * f;
  g[0] = sqrt(sp * (sp - sa) * (sp - sb) * (sp - sc));
  g[1] = nx / fa; g[2] = ny / fa; g[3] = nz / fa;
  g[4] = (X[0] + X[1] + X[2]) / 3.0;
  g[5] = (Y[0] + Y[1] + Y[2]) / 3.0;
  g[6] = (Z[0] + Z[1] + Z[2]) / 3.0;
}

// src/Mesh/DerivedData.cpp:1436-1491: volume triple(ba,ca,da)/6 and centroid
__global__ void k_geoelem(const uint64_t* __restrict__ inpoel, size_t nelem, const double* __restrict__ x,
                          const double* __restrict__ y, const double* __restrict__ z,
                          double* __restrict__ geoElem)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nelem) return;
  const uint64_t A = inpoel[4*e], B = inpoel[4*e+1], C = inpoel[4*e+2], D = inpoel[4*e+3];
  const double ba[3] = { x[B]-x[A], y[B]-y[A], z[B]-z[A] };
  const double ca[3] = { x[C]-x[A], y[C]-y[A], z[C]-z[A] };
  const double da[3] = { x[D]-x[A], y[D]-y[A], z[D]-z[A] };
  const double cx = ca[1] * da[2] - ca[2] * da[1];
  const double cy = ca[2] * da[0] - ca[0] * da[2];
  const double cz = ca[0] * da[1] - ca[1] * da[0];
  geoElem[4*e]   = (ba[0] * cx + ba[1] * cy + ba[2] * cz) / 6.0;
  geoElem[4*e+1] = (x[A] + x[B] + x[C] + x[D]) / 4.0;
  geoElem[4*e+2] = (y[A] + y[B] + y[C] + y[D]) / 4.0;
  geoElem[4*e+3] = (z[A] + z[B] + z[C] + z[D]) / 4.0;
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" int qdg_dev_facedata(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                                const double* x, const double* y, const double* z, size_t nbfac,
                                const size_t* triinpoel, int* esuel, size_t* nipfac_out,
                                size_t* inpofa, int* esuf, size_t* belem, double* geoFace,
                                double* geoElem)
{
  QDG_TRY
  if (!ctx || !inpoel || !x || !y || !z || !esuel || !nipfac_out || !inpofa || !esuf || !geoFace || !geoElem)
    return fail("qdg_dev_facedata: null argument");
  if (nbfac > 0 && (!triinpoel || !belem)) return fail("qdg_dev_facedata: null boundary arrays");
  if (nelem == 0) return fail("qdg_dev_facedata: empty mesh");
  if (nelem > (size_t)INT32_MAX / 4 || nnode > (size_t)INT32_MAX)
    return fail("qdg_dev_facedata: chunk too large for 32-bit ids");
  for (size_t i = 0; i < 4 * nelem; ++i)
    if (inpoel[i] >= nnode) return fail("qdg_dev_facedata: inpoel entry out of range");
  for (size_t i = 0; i < 3 * nbfac; ++i)
    if (triinpoel[i] >= nnode) return fail("qdg_dev_facedata: triinpoel entry out of range");
  DHIP(hipSetDevice(ctx_device(ctx)));
  hipStream_t s = ctx_stream(ctx);
  const size_t n4 = 4 * nelem, nfmax = nbfac + 2 * nelem;

  Buf<uint64_t> d_inpoel, d_tri, d_inpofa, d_belem;
  Buf<double> d_x, d_y, d_z, d_geoFace, d_geoElem;
  Buf<uint32_t> ka, kb, kc, perm, perm2, key, key2;
  Buf<int> d_esuel, d_flag, d_pos, d_esuf, d_err;
  DHIP(d_inpoel.alloc(n4)); DHIP(d_tri.alloc(3 * nbfac));
  DHIP(d_x.alloc(nnode)); DHIP(d_y.alloc(nnode)); DHIP(d_z.alloc(nnode));
  DHIP(ka.alloc(n4)); DHIP(kb.alloc(n4)); DHIP(kc.alloc(n4));
  DHIP(perm.alloc(n4)); DHIP(perm2.alloc(n4)); DHIP(key.alloc(n4)); DHIP(key2.alloc(n4));
  DHIP(d_esuel.alloc(n4)); DHIP(d_flag.alloc(n4 + 1)); DHIP(d_pos.alloc(n4 + 1)); DHIP(d_err.alloc(1));
  static_assert(sizeof(size_t) == sizeof(uint64_t), "size_t is 64 bits in this ABI");
  DHIP(hipMemcpyAsync(d_inpoel.p, inpoel, n4 * 8, hipMemcpyHostToDevice, s));
  if (nbfac) DHIP(hipMemcpyAsync(d_tri.p, triinpoel, 3 * nbfac * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(d_x.p, x, nnode * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(d_y.p, y, nnode * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(d_z.p, z, nnode * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));

  // ---- sort the 4*nelem faces by (a, b, c), ties in (element, local face) order ----
  k_face_keys<<<nblk(n4), 256, 0, s>>>(d_inpoel.p, n4, ka.p, kb.p, kc.p, perm.p);
  unsigned bits = 1;
  while (bits < 32 && ((size_t)1 << bits) < nnode) ++bits;
  size_t tmp_bytes = 0;
  DHIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key.p, key2.p, perm.p, perm2.p, n4, 0, bits, s));
  Buf<char> tmp;
  DHIP(tmp.alloc(tmp_bytes));
  const uint32_t* pass[3] = { kc.p, kb.p, ka.p };      // least significant first; the sort is stable
  uint32_t *pin = perm.p, *pout = perm2.p;
  for (int ps = 0; ps < 3; ++ps) {
    k_gather<<<nblk(n4), 256, 0, s>>>(pass[ps], pin, n4, key.p);
    DHIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, key.p, key2.p, pin, pout, n4, 0, bits, s));
    std::swap(pin, pout);
  }
  const uint32_t* sperm = pin;                          // sorted position -> 4*e + f
  // sorted keys (ka/kb/kc are reused as sa/sb/sc through key buffers)
  Buf<uint32_t> sa, sb, sc;
  DHIP(sa.alloc(n4)); DHIP(sb.alloc(n4)); DHIP(sc.alloc(n4));
  k_gather<<<nblk(n4), 256, 0, s>>>(ka.p, sperm, n4, sa.p);
  k_gather<<<nblk(n4), 256, 0, s>>>(kb.p, sperm, n4, sb.p);
  k_gather<<<nblk(n4), 256, 0, s>>>(kc.p, sperm, n4, sc.p);

  // ---- esuel ------------------------------------------------------------------------
  k_match<<<nblk(n4), 256, 0, s>>>(sa.p, sb.p, sc.p, sperm, n4, d_esuel.p, d_err.p);

  // ---- interior faces in the reference's order ---------------------------------------
  k_flag<<<nblk(n4), 256, 0, s>>>(d_esuel.p, n4, d_flag.p);
  size_t scan_bytes = 0;
  DHIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_flag.p, d_pos.p, 0, n4, rocprim::plus<int>(), s));
  Buf<char> tmp2;
  DHIP(tmp2.alloc(scan_bytes));
  DHIP(rocprim::exclusive_scan(tmp2.p, scan_bytes, d_flag.p, d_pos.p, 0, n4, rocprim::plus<int>(), s));
  int last_pos = 0, last_flag = 0, herr = 0;
  DHIP(hipMemcpyAsync(&last_pos, d_pos.p + (n4 - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(&last_flag, d_flag.p + (n4 - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  if (herr == 1) return fail("qdg_dev_facedata: face shared by more than two tets (non-manifold mesh)");
  const size_t nint = (size_t)last_pos + (size_t)last_flag, nipfac = nbfac + nint;
  if (nipfac > nfmax) return fail("qdg_dev_facedata: inconsistent face count");
  DHIP(d_inpofa.alloc(3 * nipfac)); DHIP(d_esuf.alloc(2 * nipfac)); DHIP(d_belem.alloc(nbfac));
  DHIP(d_geoFace.alloc(7 * nipfac)); DHIP(d_geoElem.alloc(4 * nelem));
  k_interior_faces<<<nblk(n4), 256, 0, s>>>(d_inpoel.p, d_esuel.p, d_flag.p, d_pos.p, n4, nbfac,
                                           d_inpofa.p, d_esuf.p);
  if (nbfac)
    k_boundary_faces<<<nblk(nbfac), 256, 0, s>>>(d_tri.p, nbfac, sa.p, sb.p, sc.p, sperm, n4, d_inpofa.p,
                                                 d_esuf.p, d_belem.p, d_err.p);
  // ---- geometry ----------------------------------------------------------------------
  k_geoface<<<nblk(nipfac), 256, 0, s>>>(d_inpofa.p, nipfac, d_x.p, d_y.p, d_z.p, d_geoFace.p);
  k_geoelem<<<nblk(nelem), 256, 0, s>>>(d_inpoel.p, nelem, d_x.p, d_y.p, d_z.p, d_geoElem.p);
  DHIP(hipGetLastError());

  DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(esuel, d_esuel.p, n4 * sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(inpofa, d_inpofa.p, 3 * nipfac * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(esuf, d_esuf.p, 2 * nipfac * sizeof(int), hipMemcpyDeviceToHost, s));
  if (nbfac) DHIP(hipMemcpyAsync(belem, d_belem.p, nbfac * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(geoFace, d_geoFace.p, 7 * nipfac * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(geoElem, d_geoElem.p, 4 * nelem * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  if (herr == 2) return fail("qdg_dev_facedata: a boundary face is not a face of any tet");
  *nipfac_out = nipfac;
  return 0;
  QDG_CATCH
}

extern "C" int qdg_mesh_from_connectivity(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                                          const double* x, const double* y, const double* z,
                                          size_t ntri, const size_t* tri, const int32_t* tri_set,
                                          qdg_mesh** out)
{
  QDG_TRY
  if (!ctx || !out || !inpoel || !x || !y || !z) return fail("qdg_mesh_from_connectivity: null argument");
  if (ntri > 0 && (!tri || !tri_set)) return fail("qdg_mesh_from_connectivity: null side-set arrays");
  *out = nullptr;
  // boundary faces in the reference's order (Partitioner.cpp:357-393), grouped by side set
  std::vector<size_t> triinpoel(3 * std::max<size_t>(ntri, 1));
  std::vector<int32_t> fset(std::max<size_t>(ntri, 1));
  size_t nbfac = 0;
  if (ntri)
    if (int rc = qdg_bnd_faces(nelem, inpoel, ntri, tri, tri_set, &nbfac, triinpoel.data(), fset.data()))
      return rc;
  const size_t nfmax = nbfac + 2 * nelem;
  std::vector<int> esuel(4 * nelem), esuf(2 * nfmax);
  std::vector<size_t> inpofa(3 * nfmax), belem(std::max<size_t>(nbfac, 1));
  std::vector<double> geoFace(7 * nfmax), geoElem(4 * nelem);
  size_t nipfac = 0;
  if (int rc = qdg_dev_facedata(ctx, nelem, nnode, inpoel, x, y, z, nbfac, triinpoel.data(), esuel.data(),
                                &nipfac, inpofa.data(), esuf.data(), belem.data(), geoFace.data(),
                                geoElem.data()))
    return rc;
  // FaceData::m_bface: side set id -> boundary face ids (faces come grouped by ascending id)
  std::vector<int32_t> ids; std::vector<size_t> off{ 0 }, faces(std::max<size_t>(nbfac, 1));
  for (size_t f = 0; f < nbfac; ++f) {
    if (ids.empty() || ids.back() != fset[f]) { if (!ids.empty()) off.push_back(f); ids.push_back(fset[f]); }
    faces[f] = f;
  }
  off.push_back(nbfac);
  if (ids.empty()) { ids.push_back(0); off.assign({ 0, 0 }); }
  qdg_bface bf{ nbfac ? ids.size() : 0, ids.data(), off.data(), faces.data() };
  return qdg_mesh_upload(ctx, nelem, nelem, nnode, inpoel, x, y, z, nbfac, nipfac, esuf.data(),
                         esuel.data(), inpofa.data(), geoFace.data(), geoElem.data(), &bf, out);
  QDG_CATCH
}
