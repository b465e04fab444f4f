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
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>

#include <rocprim/device/device_reduce.hpp>

#include "../../include/qdg.h"
#include "qdg_host.hpp"
#include "qdg_handles.hpp"

namespace qdg {
// defined in qdg_api.cpp
int ctx_device(const qdg_ctx* ctx);
hipStream_t ctx_stream(const qdg_ctx* ctx);
// defined in qdg_kernels.hip
void launch_task_geo(size_t nslot, const int* task_a, const int* task_f, const double* fgeo, double* tgeo,
                     hipStream_t s);
}  // namespace qdg

using namespace qdg;

namespace {

#define DHIP(call)                                                                \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess)                                                         \
      return ::qdg::fail(std::string(#call) + ": " + hipGetErrorString(e_));      \
  } while (0)

#define HIPCHK(call) DHIP(call)

template <class T> struct Buf {
  T* p = nullptr;
  Buf() = default;
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  ~Buf() { if (p) qdg::dev_free(p); }
  hipError_t alloc(size_t n)
  {
    if (p) { qdg::dev_free(p); p = nullptr; }
    return qdg::dev_alloc((void**)&p, (n ? n : 1) * sizeof(T));
  }
  void take(Buf& o) { if (p) qdg::dev_free(p); p = o.p; o.p = nullptr; }
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
  if (eq_next && eq_prev) atomicOr(err, 1);         // a face shared by more than two tets (error bits: 1, 2, 4)
  int v = -1;
  if (eq_next) v = (int)(perm[i + 1] >> 2);
  else if (eq_prev) v = (int)(perm[i - 1] >> 2);
  esuel[perm[i]] = v;
}

// interior and chare-boundary faces: listed by their left (lower-id) tet, which must be an owned
// one -- a face between two ghosts is none of this chunk's business
__global__ void k_flag(const int* __restrict__ esuel, size_t n4, size_t nie, int* __restrict__ flag)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int j = esuel[i];
  flag[i] = (j != -1 && (i >> 2) < nie && (int)(i >> 2) < j) ? 1 : 0;
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

// Orientation by GLOBAL tet id (context option "orient_by_gid", default on, when the chunk comes with
// its tets' global ids): the stored left tet of an interior or chare-boundary face is the one with
// the lower global id -- the rule of the serial run of the whole mesh (src/Mesh/DerivedData.cpp:
// 1127-1139: a face is kept by its lower-numbered tet) instead of the chare-local one (left = the
// owned tet, src/Inciter/DG.cpp:480-483).  HLLC's ladder falls through to the STORED right state
// when a wave speed is NaN (src/PDE/Integrate/Riemann/HLLC.hpp:93-124), so only with a
// partition-independent orientation is a partitioned run equal to the serial one at such faces.
// The face's nodes are taken in the new left tet's local face order, as the serial run would store them.
__global__ void k_orient_gid(const uint64_t* __restrict__ inpoel, const uint64_t* __restrict__ gid, size_t nbfac,
                             size_t nipfac, uint64_t* __restrict__ inpofa, int* __restrict__ esuf)
{
  const size_t f = nbfac + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nipfac) return;
  const int el = esuf[2 * f], er = esuf[2 * f + 1];
  if (er < 0 || gid[el] < gid[er]) return;
  const uint64_t a = inpofa[3 * f], b = inpofa[3 * f + 1], c = inpofa[3 * f + 2];
  int q = 0;                                         // local face of er = its node that is not on the face
  for (int k = 0; k < 4; ++k) {
    const uint64_t g = inpoel[4 * (size_t)er + k];
    if (g != a && g != b && g != c) q = k;
  }
  for (int j = 0; j < 3; ++j) inpofa[3 * f + j] = inpoel[4 * (size_t)er + c_lpofa[q][j]];
  esuf[2 * f] = er; esuf[2 * f + 1] = el;
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
  if (lo >= n || sa[lo] != k0 || sb[lo] != k1 || sc[lo] != k2) { atomicOr(err, 2); return; }
  const uint64_t e = perm[lo] >> 2;
  belem[f] = e;
  esuf[2 * f] = (int)e;
  esuf[2 * f + 1] = -1;
}

// boundary faces whose tet is known (regenerated by dev_bnd_faces: fd.belem filled there)
__global__ void k_boundary_faces_known(const uint64_t* __restrict__ tri, const uint64_t* __restrict__ belem,
                                       size_t nbfac, uint64_t* __restrict__ inpofa, int* __restrict__ esuf)
{
  const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nbfac) return;
  inpofa[3 * f] = tri[3 * f]; inpofa[3 * f + 1] = tri[3 * f + 1]; inpofa[3 * f + 2] = tri[3 * f + 2];
  esuf[2 * f] = (int)belem[f];
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
                          double* __restrict__ geoElem, int* __restrict__ err)
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
  const double vol = (ba[0] * cx + ba[1] * cy + ba[2] * cz) / 6.0;
  if (!(vol > 0.0)) atomicOr(err, 4);           // the reference asserts a positive Jacobian (DerivedData.cpp:1478-1480)
  geoElem[4*e]   = vol;
  geoElem[4*e+1] = (x[A] + x[B] + x[C] + x[D]) / 4.0;
  geoElem[4*e+2] = (y[A] + y[B] + y[C] + y[D]) / 4.0;
  geoElem[4*e+3] = (z[A] + z[B] + z[C] + z[D]) / 4.0;
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

// the FaceData arrays and the geometry of a chunk, resident on the device
struct DevFD {
  Buf<uint64_t> inpoel, tri, inpofa, belem;
  Buf<uint64_t> gid;         // global tet ids of the chunk's tets (when the caller gave them)
  bool orient = false;       // ... and faces are oriented by them (option orient_by_gid)
  Buf<double> x, y, z, geoFace, geoElem;
  Buf<int> esuel, esuf;
  size_t nelem = 0, nnode = 0, nbfac = 0, nipfac = 0;
  size_t nie = 0;     // owned tets [0, nie); ghosts [nie, nelem) (nie == nelem: a chunk without ghosts)
  bool nonpos_vol = false;   // some tet has a non-positive volume (an error for a mesh to compute on)
  bool belem_known = false;  // fd.belem filled with the boundary faces' tets (dev_bnd_faces)
  std::vector<int32_t> fset; // side-set id of every boundary face (dev_bnd_faces)
};

// the sorted face keys of a chunk: what esuel is derived from in the general build, and what the tet of a
// caller-supplied boundary triangle is looked up in (qdg_dev_facedata)
struct SortedFaces {
  Buf<uint32_t> sa, sb, sc, perm, perm2;
  const uint32_t* sperm = nullptr;          // sorted position -> 4 * tet + local face
};

// connectivity and coordinates of a chunk to the device (validated on the host first)
static int dev_upload_mesh(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel, const double* x,
                           const double* y, const double* z, DevFD& fd)
{
  if (!ctx || !inpoel || !x || !y || !z) return fail("qdg_dev_facedata: null argument");
  if (nelem == 0) return fail("qdg_dev_facedata: empty mesh");
  if (nelem > (size_t)INT32_MAX / 4 || nnode > (size_t)INT32_MAX)
    return fail("qdg_dev_facedata: chunk too large for 32-bit ids");
  for (size_t i = 0; i < 4 * nelem; ++i)
    if (inpoel[i] >= nnode) return fail("qdg_dev_facedata: inpoel entry out of range");
  DHIP(hipSetDevice(ctx_device(ctx)));
  hipStream_t s = ctx_stream(ctx);
  fd.nelem = nelem; fd.nnode = nnode; fd.nie = nelem;
  static_assert(sizeof(size_t) == sizeof(uint64_t), "size_t is 64 bits in this ABI");
  DHIP(fd.inpoel.alloc(4 * nelem));
  DHIP(fd.x.alloc(nnode)); DHIP(fd.y.alloc(nnode)); DHIP(fd.z.alloc(nnode));
  DHIP(hipMemcpyAsync(fd.inpoel.p, inpoel, 4 * nelem * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(fd.x.p, x, nnode * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(fd.y.p, y, nnode * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(fd.z.p, z, nnode * 8, hipMemcpyHostToDevice, s));
  return 0;
}

// Boundary-face regeneration of the mesh loader (src/Inciter/Partitioner.cpp:357-393) on the
// device: the side-set triangles are order-independent keys; every tet, in order, contributes
// its faces {0,2,1},{0,1,3},{0,3,2},{1,2,3} that match a key, in that node order; faces come out
// grouped by ascending side-set id, within a set in tet order (same as qdg_bnd_faces).
__constant__ int c_bfa[4][3] = { { 0, 2, 1 }, { 0, 1, 3 }, { 0, 3, 2 }, { 1, 2, 3 } };

__global__ void k_bnd_match(const uint64_t* __restrict__ inpoel, const int* __restrict__ esuel, size_t n4,
                            const uint32_t* __restrict__ ta, const uint32_t* __restrict__ tb,
                            const uint32_t* __restrict__ tc, const uint32_t* __restrict__ trank, size_t ntri,
                            size_t nie, uint64_t* __restrict__ okey, unsigned long long* __restrict__ count)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const size_t e = i >> 2; const int f = (int)(i & 3);
  okey[i] = ~0ull;
  if (e >= nie) return;                                // a ghost's boundary faces belong to its owner
  // c_bfa[f] is the face opposite node 3 - f, i.e. local face 3 - f of esuel's numbering (c_lpofa):
  // only a FREE face can be a physical-boundary face
  if (esuel[4 * e + (3 - f)] != -1) return;
  uint32_t k0 = (uint32_t)inpoel[4 * e + c_bfa[f][0]], k1 = (uint32_t)inpoel[4 * e + c_bfa[f][1]],
           k2 = (uint32_t)inpoel[4 * e + c_bfa[f][2]], t;
  if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
  if (k1 > k2) { t = k1; k1 = k2; k2 = t; }
  if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
  size_t lo = 0, hi = ntri;
  while (lo < hi) {
    const size_t mid = (lo + hi) >> 1;
    const bool less = ta[mid] < k0 || (ta[mid] == k0 && (tb[mid] < k1 || (tb[mid] == k1 && tc[mid] < k2)));
    if (less) lo = mid + 1; else hi = mid;
  }
  const bool hit = lo < ntri && ta[lo] == k0 && tb[lo] == k1 && tc[lo] == k2;
  if (hit) { okey[i] = ((uint64_t)trank[lo] << 40) | (uint64_t)i; atomicAdd(count, 1ull); }
}

struct NotSentinel {
  __host__ __device__ bool operator()(const uint64_t& k) const { return k != ~0ull; }
};

__global__ void k_bnd_emit(const uint64_t* __restrict__ skey, size_t nb, const uint64_t* __restrict__ inpoel,
                           uint64_t* __restrict__ tri, uint64_t* __restrict__ belem, int* __restrict__ rank_of_face)
{
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const uint64_t i = skey[b] & 0xffffffffffull;
  const size_t e = i >> 2; const int f = (int)(i & 3);
  tri[3 * b] = inpoel[4 * e + c_bfa[f][0]];
  tri[3 * b + 1] = inpoel[4 * e + c_bfa[f][1]];
  tri[3 * b + 2] = inpoel[4 * e + c_bfa[f][2]];
  belem[b] = e;
  rank_of_face[b] = (int)(skey[b] >> 40);
}

// Boundary faces of a chunk from the SORTED side-set triangle keys on the device (ta <= tb <= tc per
// triangle, triangles in lexicographic order, trank = rank of the triangle's side set in `sets`) and the
// chunk's esuel: the free faces of the owned tets are looked up in the keys, the hits compacted and sorted
// by (set rank, tet, face) -- a sort over the boundary faces, not over all 4 * nelem faces.
// out: fd.tri (the faces in the loader's node order), fd.belem, fd.nbfac, fd.fset
static int dev_bnd_faces_core(qdg_ctx* ctx, DevFD& fd, size_t ntri, const uint32_t* ta, const uint32_t* tb,
                              const uint32_t* tc, const uint32_t* tr, const std::vector<int32_t>& sets)
{
  hipStream_t s = ctx_stream(ctx);
  const size_t n4 = 4 * fd.nelem;
  Buf<uint64_t> okey, ckey, skey;
  Buf<unsigned long long> cnt;
  Buf<size_t> cnt2;
  DHIP(okey.alloc(n4)); DHIP(cnt.alloc(1)); DHIP(cnt2.alloc(1));
  DHIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s));
  k_bnd_match<<<nblk(n4), 256, 0, s>>>(fd.inpoel.p, fd.esuel.p, n4, ta, tb, tc, tr, ntri, fd.nie, okey.p, cnt.p);
  unsigned long long hits = 0;
  DHIP(hipMemcpyAsync(&hits, cnt.p, sizeof hits, hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  const size_t nb = (size_t)hits;
  // the hits are a surface quantity: compact them, then sort by (set rank, tet, face)
  DHIP(ckey.alloc(nb));
  if (nb) {
    size_t bytes = 0;
    DHIP(rocprim::select(nullptr, bytes, okey.p, ckey.p, cnt2.p, n4, NotSentinel(), s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    DHIP(rocprim::select(tmp.p, bytes, okey.p, ckey.p, cnt2.p, n4, NotSentinel(), s));
    DHIP(hipStreamSynchronize(s));
  }
  fd.nbfac = nb;
  fd.fset.assign(nb, 0);
  DHIP(fd.tri.alloc(3 * nb)); DHIP(fd.belem.alloc(nb));
  fd.belem_known = true;
  if (!nb) return 0;
  DHIP(skey.alloc(nb));
  {
    size_t bytes = 0;
    DHIP(rocprim::radix_sort_keys(nullptr, bytes, ckey.p, skey.p, nb, 0, 64, s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    DHIP(rocprim::radix_sort_keys(tmp.p, bytes, ckey.p, skey.p, nb, 0, 64, s));
    DHIP(hipStreamSynchronize(s));
  }
  Buf<int> rk;
  DHIP(rk.alloc(nb));
  k_bnd_emit<<<nblk(nb), 256, 0, s>>>(skey.p, nb, fd.inpoel.p, fd.tri.p, fd.belem.p, rk.p);
  std::vector<int> hrk(nb);
  DHIP(hipMemcpyAsync(hrk.data(), rk.p, nb * sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  for (size_t b = 0; b < nb; ++b) fd.fset[b] = sets[hrk[b]];
  return 0;
}

// fd.inpoel and fd.esuel resident; side-set triangles from the host (any node order, tagged with their set)
static int dev_bnd_faces(qdg_ctx* ctx, DevFD& fd, size_t ntri, const size_t* tri, const int32_t* tri_set)
{
  hipStream_t s = ctx_stream(ctx);
  fd.nbfac = 0;
  fd.fset.clear();
  fd.belem_known = true;
  if (ntri == 0) { DHIP(fd.tri.alloc(1)); DHIP(fd.belem.alloc(1)); return 0; }
  // sorted keys of the side-set triangles (host: ntri is a surface quantity)
  struct K { uint32_t a, b, c; int32_t set; };
  std::vector<K> keys(ntri);
  std::vector<int32_t> sets;
  for (size_t i = 0; i < ntri; ++i) {
    size_t a = tri[3 * i], b = tri[3 * i + 1], c = tri[3 * i + 2];
    if (a >= fd.nnode || b >= fd.nnode || c >= fd.nnode) return fail("qdg_mesh_from_connectivity: side-set triangle node out of range");
    if (a > b) std::swap(a, b);
    if (b > c) std::swap(b, c);
    if (a > b) std::swap(a, b);
    keys[i] = { (uint32_t)a, (uint32_t)b, (uint32_t)c, tri_set[i] };
    sets.push_back(tri_set[i]);
  }
  std::sort(keys.begin(), keys.end(), [](const K& p, const K& q) {
    return p.a != q.a ? p.a < q.a : p.b != q.b ? p.b < q.b : p.c != q.c ? p.c < q.c : p.set < q.set; });
  // A triangle listed in several side sets: the mesh loader's map from triangle to side set is filled set by
  // set in ascending id (`faceside[tri] = s.first` over the std::map m_bface, src/Inciter/Partitioner.cpp:358-364,
  // Partitioner.hpp:198), so the LAST set that lists a triangle owns it and the tet face is integrated once, with
  // that set's condition (Boundary.cpp:84-86 then finds it in that set only).  The keys are sorted by (nodes,
  // set): the last entry of every run of equal node triples stays (as qdg_bnd_faces does on the host).
  {
    size_t w = 0;
    for (size_t i = 0; i < ntri; ++i) {
      const bool last = i + 1 == ntri || keys[i].a != keys[i + 1].a || keys[i].b != keys[i + 1].b || keys[i].c != keys[i + 1].c;
      if (last) keys[w++] = keys[i];
    }
    ntri = w;
    keys.resize(ntri);
  }
  std::sort(sets.begin(), sets.end());
  sets.erase(std::unique(sets.begin(), sets.end()), sets.end());
  if (sets.size() >= (1u << 20)) return fail("qdg_mesh_from_connectivity: too many side sets");
  std::vector<uint32_t> ha(ntri), hb(ntri), hc(ntri), hr(ntri);
  for (size_t i = 0; i < ntri; ++i) {
    ha[i] = keys[i].a; hb[i] = keys[i].b; hc[i] = keys[i].c;
    hr[i] = (uint32_t)(std::lower_bound(sets.begin(), sets.end(), keys[i].set) - sets.begin());
  }
  Buf<uint32_t> ta, tb, tc, tr;
  DHIP(ta.alloc(ntri)); DHIP(tb.alloc(ntri)); DHIP(tc.alloc(ntri)); DHIP(tr.alloc(ntri));
  DHIP(hipMemcpyAsync(ta.p, ha.data(), ntri * 4, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(tb.p, hb.data(), ntri * 4, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(tc.p, hc.data(), ntri * 4, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(tr.p, hr.data(), ntri * 4, hipMemcpyHostToDevice, s));
  return dev_bnd_faces_core(ctx, fd, ntri, ta.p, tb.p, tc.p, tr.p, sets);     // (synchronises the stream)
}

// part A of the FaceData build: esuel from the sorted face keys (3 stable radix sorts over 4 * nelem faces)
static int dev_esuel_by_sort(qdg_ctx* ctx, DevFD& fd, SortedFaces& sf)
{
  DHIP(hipSetDevice(ctx_device(ctx)));
  hipStream_t s = ctx_stream(ctx);
  const size_t nelem = fd.nelem, nnode = fd.nnode;
  const size_t n4 = 4 * nelem;
  Buf<uint32_t> ka, kb, kc, key, key2;
  Buf<int> d_err;
  DHIP(ka.alloc(n4)); DHIP(kb.alloc(n4)); DHIP(kc.alloc(n4));
  DHIP(sf.perm.alloc(n4)); DHIP(sf.perm2.alloc(n4)); DHIP(key.alloc(n4)); DHIP(key2.alloc(n4));
  DHIP(fd.esuel.alloc(n4)); DHIP(d_err.alloc(1));
  DHIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));
  // ---- sort the 4*nelem faces by (a, b, c), ties in (element, local face) order ----
  k_face_keys<<<nblk(n4), 256, 0, s>>>(fd.inpoel.p, n4, ka.p, kb.p, kc.p, sf.perm.p);
  unsigned bits = 1;
  while (bits < 32 && ((size_t)1 << bits) < nnode) ++bits;
  size_t tmp_bytes = 0;
  DHIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, key.p, key2.p, sf.perm.p, sf.perm2.p, n4, 0, bits, s));
  Buf<char> tmp;
  DHIP(tmp.alloc(tmp_bytes));
  const uint32_t* pass[3] = { kc.p, kb.p, ka.p };      // least significant first; the sort is stable
  uint32_t *pin = sf.perm.p, *pout = sf.perm2.p;
  for (int ps = 0; ps < 3; ++ps) {
    k_gather<<<nblk(n4), 256, 0, s>>>(pass[ps], pin, n4, key.p);
    DHIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, key.p, key2.p, pin, pout, n4, 0, bits, s));
    std::swap(pin, pout);
  }
  sf.sperm = pin;                                       // sorted position -> 4*e + f
  DHIP(sf.sa.alloc(n4)); DHIP(sf.sb.alloc(n4)); DHIP(sf.sc.alloc(n4));
  k_gather<<<nblk(n4), 256, 0, s>>>(ka.p, sf.sperm, n4, sf.sa.p);
  k_gather<<<nblk(n4), 256, 0, s>>>(kb.p, sf.sperm, n4, sf.sb.p);
  k_gather<<<nblk(n4), 256, 0, s>>>(kc.p, sf.sperm, n4, sf.sc.p);
  k_match<<<nblk(n4), 256, 0, s>>>(sf.sa.p, sf.sb.p, sf.sc.p, sf.sperm, n4, fd.esuel.p, d_err.p);
  int herr = 0;
  DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));                        // (also: the temporaries above are released on return)
  if (herr & 1) return fail("qdg_dev_facedata: face shared by more than two tets (non-manifold mesh)");
  return 0;
}

// part B: interior / chare-boundary faces in the reference's order, boundary faces, geometry -- from
// fd.esuel and the boundary faces fd.tri (with their tets in fd.belem when fd.belem_known, else looked
// up in the sorted face keys sf)
static int dev_faces_geometry(qdg_ctx* ctx, DevFD& fd, const SortedFaces* sf)
{
  DHIP(hipSetDevice(ctx_device(ctx)));
  hipStream_t s = ctx_stream(ctx);
  const size_t nelem = fd.nelem, nbfac = fd.nbfac;
  const size_t n4 = 4 * nelem, nfmax = nbfac + 2 * nelem;
  if (!fd.belem_known && !sf) return fail("qdg_dev_facedata: internal: boundary faces without tets or keys");
  Buf<int> d_flag, d_pos, d_err;
  DHIP(d_flag.alloc(n4 + 1)); DHIP(d_pos.alloc(n4 + 1)); DHIP(d_err.alloc(1));
  DHIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));
  k_flag<<<nblk(n4), 256, 0, s>>>(fd.esuel.p, n4, fd.nie, d_flag.p);
  size_t scan_bytes = 0;
  DHIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_flag.p, d_pos.p, 0, n4, rocprim::plus<int>(), s));
  Buf<char> tmp2;
  DHIP(tmp2.alloc(scan_bytes));
  DHIP(rocprim::exclusive_scan(tmp2.p, scan_bytes, d_flag.p, d_pos.p, 0, n4, rocprim::plus<int>(), s));
  int last_pos = 0, last_flag = 0, herr = 0;
  DHIP(hipMemcpyAsync(&last_pos, d_pos.p + (n4 - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(&last_flag, d_flag.p + (n4 - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  const size_t nint = (size_t)last_pos + (size_t)last_flag, nipfac = nbfac + nint;
  if (nipfac > nfmax) return fail("qdg_dev_facedata: inconsistent face count");
  fd.nipfac = nipfac;
  DHIP(fd.inpofa.alloc(3 * nipfac)); DHIP(fd.esuf.alloc(2 * nipfac));
  if (!fd.belem_known) DHIP(fd.belem.alloc(nbfac));
  DHIP(fd.geoFace.alloc(7 * nipfac)); DHIP(fd.geoElem.alloc(4 * nelem));
  k_interior_faces<<<nblk(n4), 256, 0, s>>>(fd.inpoel.p, fd.esuel.p, d_flag.p, d_pos.p, n4, nbfac,
                                           fd.inpofa.p, fd.esuf.p);
  if (fd.gid.p && fd.orient && nint)
    k_orient_gid<<<nblk(nint), 256, 0, s>>>(fd.inpoel.p, fd.gid.p, nbfac, nipfac, fd.inpofa.p, fd.esuf.p);
  if (nbfac) {
    if (fd.belem_known)
      k_boundary_faces_known<<<nblk(nbfac), 256, 0, s>>>(fd.tri.p, fd.belem.p, nbfac, fd.inpofa.p, fd.esuf.p);
    else
      k_boundary_faces<<<nblk(nbfac), 256, 0, s>>>(fd.tri.p, nbfac, sf->sa.p, sf->sb.p, sf->sc.p, sf->sperm, n4,
                                                   fd.inpofa.p, fd.esuf.p, fd.belem.p, d_err.p);
  }
  // ---- geometry ----------------------------------------------------------------------
  k_geoface<<<nblk(nipfac), 256, 0, s>>>(fd.inpofa.p, nipfac, fd.x.p, fd.y.p, fd.z.p, fd.geoFace.p);
  k_geoelem<<<nblk(nelem), 256, 0, s>>>(fd.inpoel.p, nelem, fd.x.p, fd.y.p, fd.z.p, fd.geoElem.p, d_err.p);
  DHIP(hipGetLastError());
  DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  // (qdg_dev_facedata itself is a statement about connectivity: the reference's derived-data unit meshes
  // are not all positively oriented; the mesh BUILD refuses such a tet)
  fd.nonpos_vol = (herr & 4) != 0;      // its own bit: an inverted tet must not hide an unmatched boundary face
  if (herr & 2) return fail("qdg_dev_facedata: a boundary face is not a face of any tet");
  return 0;
}

// FaceData + geometry from the resident connectivity, coordinates and caller-supplied boundary faces (fd.tri)
static int dev_facedata_from(qdg_ctx* ctx, DevFD& fd)
{
  SortedFaces sf;
  if (int rc = dev_esuel_by_sort(ctx, fd, sf)) return rc;
  fd.belem_known = false;
  return dev_faces_geometry(ctx, fd, &sf);
}

static int dev_facedata_keep(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                             const double* x, const double* y, const double* z, size_t nbfac,
                             const size_t* triinpoel, DevFD& fd)
{
  if (nbfac > 0 && !triinpoel) return fail("qdg_dev_facedata: null boundary arrays");
  for (size_t i = 0; i < 3 * nbfac; ++i)
    if (triinpoel[i] >= nnode) return fail("qdg_dev_facedata: triinpoel entry out of range");
  if (int rc = dev_upload_mesh(ctx, nelem, nnode, inpoel, x, y, z, fd)) return rc;
  fd.nbfac = nbfac;
  DHIP(fd.tri.alloc(3 * nbfac));
  if (nbfac) DHIP(hipMemcpyAsync(fd.tri.p, triinpoel, 3 * nbfac * 8, hipMemcpyHostToDevice, ctx_stream(ctx)));
  return dev_facedata_from(ctx, fd);
}

extern "C" int qdg_dev_facedata(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                                const double* x, const double* y, const double* z, size_t nbfac,
                                const size_t* triinpoel, int* esuel, size_t* nipfac_out,
                                size_t* inpofa, int* esuf, size_t* belem, double* geoFace,
                                double* geoElem)
{
  QDG_TRY
  if (!esuel || !nipfac_out || !inpofa || !esuf || !geoFace || !geoElem) return fail("qdg_dev_facedata: null argument");
  if (nbfac > 0 && !belem) return fail("qdg_dev_facedata: null boundary arrays");
  DevFD fd;
  if (int rc = dev_facedata_keep(ctx, nelem, nnode, inpoel, x, y, z, nbfac, triinpoel, fd)) return rc;
  hipStream_t s = ctx_stream(ctx);
  const size_t n4 = 4 * nelem, nipfac = fd.nipfac;
  DHIP(hipMemcpyAsync(esuel, fd.esuel.p, n4 * sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(inpofa, fd.inpofa.p, 3 * nipfac * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(esuf, fd.esuf.p, 2 * nipfac * sizeof(int), hipMemcpyDeviceToHost, s));
  if (nbfac) DHIP(hipMemcpyAsync(belem, fd.belem.p, nbfac * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(geoFace, fd.geoFace.p, 7 * nipfac * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(geoElem, fd.geoElem.p, 4 * nelem * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  *nipfac_out = nipfac;
  return 0;
  QDG_CATCH
}

// ======================================================================================
// Device layout of a chunk WITHOUT ghosts built on the device (second step):
// everything qdg_mesh_upload derives on the host -- Morton order of the tets, node and face
// numbering by first touch, neighbour / face-code / face-id planes, packed geometry records,
// the face-task lists of the tile kernels -- from the resident FaceData (DevFD), with the SAME
// ordering rules, so that a mesh built here is row for row the mesh qdg_mesh_upload builds.
// This is what a re-mesh during time stepping pays (DG::resizePostAMR, DG.cpp:1536-1612).
namespace {

__device__ __forceinline__ unsigned long long dord(double v)     // order-preserving bits
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ inline double dord_inv(unsigned long long k)
{
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double v; std::memcpy(&v, &b, 8); return v;
}

__global__ void k_bbox(const double* __restrict__ geoElem, size_t ne, unsigned long long* __restrict__ mm)
{
  // wave reduction first, one atomic per wave and bound: every thread hitting the same six words cost 10.7 ms at
  // 10.1 M tets (86 ms of the 106 ms "Morton order" stage at 80.9 M, profiles/r04_nx119_kernel_stats.csv)
  // (a fixed grid striding over the tets: a few thousand atomics in all)
  unsigned long long lo[3] = { ~0ull, ~0ull, ~0ull }, hi[3] = { 0ull, 0ull, 0ull };
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const unsigned long long k = dord(geoElem[4 * e + 1 + d]);
      lo[d] = k < lo[d] ? k : lo[d];
      hi[d] = k > hi[d] ? k : hi[d];
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d)
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long a = __shfl_down(lo[d], off, 64), b = __shfl_down(hi[d], off, 64);
      lo[d] = a < lo[d] ? a : lo[d];
      hi[d] = b > hi[d] ? b : hi[d];
    }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { atomicMin(mm + d, lo[d]); atomicMax(mm + 3 + d, hi[d]); }
  }
}

__device__ __forceinline__ uint64_t spread21d(uint64_t v)
{
  v &= 0x1fffff;
  v = (v | v << 32) & 0x1f00000000ffffULL;
  v = (v | v << 16) & 0x1f0000ff0000ffULL;
  v = (v | v << 8) & 0x100f00f00f00f00fULL;
  v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
  v = (v | v << 2) & 0x1249249249249249ULL;
  return v;
}

// owned tets: Morton key of the centroid, bit 63 set for a tet with a ghost neighbour (those go
// last, still in curve order: launches over the leading rows never touch the halo)
// (two_hop: also the tets with a neighbour that has a ghost neighbour -- the send rows of a two-layer plan)
__global__ void k_morton(const double* __restrict__ geoElem, size_t ne, const int* __restrict__ esuel, double lx,
                         double ly, double lz, double ext, uint64_t* __restrict__ key, uint32_t* __restrict__ val,
                         int* __restrict__ ninner, int two_hop)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool inner = false;
  if (e < ne) {
    bool halo = false;
    for (int lf = 0; lf < 4; ++lf) {
      const int nb = esuel[4 * e + lf];
      halo = halo || nb >= (int)ne;
      if (two_hop && nb >= 0 && nb < (int)ne)
        for (int l2 = 0; l2 < 4; ++l2) halo = halo || esuel[4 * (size_t)nb + l2] >= (int)ne;
    }
    const double lo[3] = { lx, ly, lz };
    uint64_t k = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double t = (geoElem[4 * e + 1 + d] - lo[d]) / ext;
      const uint64_t q = (uint64_t)fmin(2097151.0, fmax(0.0, t * 2097152.0));
      k |= spread21d(q) << d;
    }
    key[e] = k | (halo ? 0x8000000000000000ull : 0ull); val[e] = (uint32_t)e;
    inner = !halo;
  }
  // one atomic per wave (every tet of a chunk without ghosts used to add 1 to the same word)
  const unsigned long long b = __ballot(inner);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(ninner, (int)__popcll(b));
}

__global__ void k_ghost_ids(uint32_t* __restrict__ val, size_t nie, size_t ne)
{
  const size_t e = nie + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ne) val[e] = (uint32_t)e;
}

__global__ void k_invert(const uint32_t* __restrict__ d2h, size_t n, int* __restrict__ h2d, int* __restrict__ d2h_i)
{
  const size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n) return;
  h2d[d2h[d]] = (int)d; d2h_i[d] = (int)d2h[d];
}

__global__ void k_fill_u32(uint32_t* p, size_t n, uint32_t v)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// (host tet, local face) -> reference face id, from esuf (one thread per face)
__global__ void k_rface(size_t nf, const int* __restrict__ esuf, const int* __restrict__ esuel,
                        const uint64_t* __restrict__ inpoel, const uint64_t* __restrict__ inpofa,
                        int* __restrict__ rface, int* __restrict__ err)
{
  const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int el = esuf[2 * f], er = esuf[2 * f + 1];
  if (er == -1) {
    int found = -1;
    for (int lf = 0; lf < 4 && found < 0; ++lf) {
      int cnt = 0;
      for (int j = 0; j < 3; ++j)
        for (int k = 0; k < 3; ++k)
          if (inpoel[4 * (size_t)el + c_lpofa[lf][j]] == inpofa[3 * f + k]) ++cnt;
      if (cnt == 3) found = lf;
    }
    if (found < 0) { *err = 3; return; }
    rface[4 * (size_t)el + found] = (int)f;
  } else {
    int a = -1, b = -1;
    for (int lf = 0; lf < 4; ++lf) if (esuel[4 * (size_t)el + lf] == er) a = lf;
    for (int lf = 0; lf < 4; ++lf) if (esuel[4 * (size_t)er + lf] == el) b = lf;
    if (a < 0 || b < 0) { *err = 4; return; }
    rface[4 * (size_t)el + a] = (int)f;
    rface[4 * (size_t)er + b] = (int)f;
  }
}

// first touch of nodes and faces in device element order: slot = 4*d + i
__global__ void k_first_touch(size_t ne, const int* __restrict__ d2h, const uint64_t* __restrict__ inpoel,
                              const int* __restrict__ rface, uint32_t* __restrict__ first_node,
                              uint32_t* __restrict__ first_face)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * ne) return;
  const size_t d = i >> 2; const int k = (int)(i & 3);
  const size_t h = (size_t)d2h[d];
  atomicMin(first_node + inpoel[4 * h + k], (uint32_t)i);
  if (rface[4 * h + k] >= 0) atomicMin(first_face + rface[4 * h + k], (uint32_t)i);   // (-1: a ghost's other faces)
}

// flags of the slots that are the first touch of their node / face
__global__ void k_touch_flags(size_t ne, const int* __restrict__ d2h, const uint64_t* __restrict__ inpoel,
                              const int* __restrict__ rface, const uint32_t* __restrict__ first_node,
                              const uint32_t* __restrict__ first_face, int* __restrict__ flagn, int* __restrict__ flagf)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * ne) return;
  const size_t d = i >> 2; const int k = (int)(i & 3);
  const size_t h = (size_t)d2h[d];
  flagn[i] = first_node[inpoel[4 * h + k]] == (uint32_t)i ? 1 : 0;
  const int f = rface[4 * h + k];
  flagf[i] = (f >= 0 && first_face[f] == (uint32_t)i) ? 1 : 0;
}

// new id = number of first-touch slots before the node's / face's own; count[0], count[1] = how many
__global__ void k_touch_assign(size_t ne, const int* __restrict__ d2h, const uint64_t* __restrict__ inpoel,
                               const int* __restrict__ rface, const int* __restrict__ flagn,
                               const int* __restrict__ flagf, const int* __restrict__ posn,
                               const int* __restrict__ posf, int* __restrict__ nnew, int* __restrict__ fmap,
                               int* __restrict__ count)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * ne) return;
  const size_t d = i >> 2; const int k = (int)(i & 3);
  const size_t h = (size_t)d2h[d];
  if (flagn[i]) nnew[inpoel[4 * h + k]] = posn[i];
  if (flagf[i]) fmap[rface[4 * h + k]] = posf[i];
  if (i == 4 * ne - 1) { count[0] = posn[i] + flagn[i]; count[1] = posf[i] + flagf[i]; }
}

__global__ void k_layout_rows(size_t ne, int stride, const int* __restrict__ d2h, const int* __restrict__ h2d,
                              const uint64_t* __restrict__ inpoel, const int* __restrict__ esuel,
                              const int* __restrict__ esuf, const int* __restrict__ rface,
                              const int* __restrict__ nnew, const int* __restrict__ fmap,
                              const int* __restrict__ bcface, const double* __restrict__ geoElem,
                              int* __restrict__ o_inpoel, int* __restrict__ o_nbr, int* __restrict__ o_finfo,
                              int* __restrict__ o_fid, double* __restrict__ o_vol, int* __restrict__ err)
{
  const size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ne) return;
  const size_t h = (size_t)d2h[d];
  for (int i = 0; i < 4; ++i) o_inpoel[(size_t)i * stride + d] = nnew[inpoel[4 * h + i]];
  o_vol[d] = geoElem[4 * h];
  for (int lf = 0; lf < 4; ++lf) {
    const int f = rface[4 * h + lf];
    if (f < 0) { *err = 6; continue; }               // free face that no side set lists
    o_fid[(size_t)lf * stride + d] = fmap[f];
    const int nb = esuel[4 * h + lf];
    int info = ((size_t)esuf[2 * f] == h) ? (1 << 6) : 0;
    if (nb < 0) {
      o_nbr[(size_t)lf * stride + d] = -(1 + bcface[f]);
    } else {
      o_nbr[(size_t)lf * stride + d] = h2d[nb];
      for (int j = 0; j < 3; ++j) {
        const uint64_t g = inpoel[4 * h + c_lpofa[lf][j]];
        int m = -1;
        for (int q = 0; q < 4; ++q) if (inpoel[4 * (size_t)nb + q] == g) m = q;
        if (m < 0) { *err = 5; m = 0; }
        info |= m << (2 * j);
      }
    }
    o_finfo[(size_t)lf * stride + d] = info;
  }
}

// ghost rows [nie, ne): their face neighbours in device numbering (-1: none in this chunk).  Read only by the
// limiter of a rank that limits its layer-1 ghosts itself (two ghost layers, qdg_halo_set_depth): every face
// neighbour of such a ghost is in the chunk, so -1 there is a physical-boundary face, which the limiters skip
// (Limiter.cpp:67-68, 212-213)
__global__ void k_layout_ghost_nbr(size_t nie, size_t ne, int stride, const int* __restrict__ h2d,
                                   const int* __restrict__ esuel, int* __restrict__ o_nbr)
{
  const size_t d = nie + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= ne) return;
  for (int lf = 0; lf < 4; ++lf) {
    const int nb = esuel[4 * d + lf];                 // (ghosts keep their rows: d2h[d] = d)
    o_nbr[(size_t)lf * stride + d] = nb >= 0 ? h2d[nb] : -1;
  }
}

__global__ void k_layout_nodes(size_t nnode, const int* __restrict__ nnew, const double* __restrict__ x,
                               const double* __restrict__ y, const double* __restrict__ z,
                               double* __restrict__ ox, double* __restrict__ oy, double* __restrict__ oz,
                               double* __restrict__ xyz4)
{
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nnode) return;
  const int k = nnew[n];
  if (k < 0) return;
  ox[k] = x[n]; oy[k] = y[n]; oz[k] = z[n];
  xyz4[4 * (size_t)k] = x[n]; xyz4[4 * (size_t)k + 1] = y[n]; xyz4[4 * (size_t)k + 2] = z[n]; xyz4[4 * (size_t)k + 3] = 0.0;
}

__global__ void k_layout_faces(size_t nf, const int* __restrict__ fmap, const double* __restrict__ geoFace,
                               double* __restrict__ area, double* __restrict__ nx, double* __restrict__ ny,
                               double* __restrict__ nz, double* __restrict__ fgeo)
{
  const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int k = fmap[f];
  if (k < 0) return;
  const double a = geoFace[7 * f], b = geoFace[7 * f + 1], c = geoFace[7 * f + 2], d = geoFace[7 * f + 3];
  area[k] = a; nx[k] = b; ny[k] = c; nz[k] = d;
  fgeo[4 * (size_t)k] = a; fgeo[4 * (size_t)k + 1] = b; fgeo[4 * (size_t)k + 2] = c; fgeo[4 * (size_t)k + 3] = d;
}

// face tasks: sort key (tile, kind, local face) of every (device row, local face); in-tile faces
// are listed by their left tet only (the other side gets the "dropped" key)
__device__ __forceinline__ bool task_of(size_t d, int lf, int stride, int tile_rows, size_t ne, const int* nbr,
                                        const int* finfo, const int* fid, int& a, int& nbid, int& f, uint32_t& key)
{
  const int nb = nbr[(size_t)lf * stride + d], info = finfo[(size_t)lf * stride + d];
  const int own_left = (info >> 6) & 1, code = info & 63;
  const size_t tile = d / tile_rows, e0 = tile * tile_rows, e1 = (e0 + tile_rows < ne) ? e0 + tile_rows : ne;
  int kind, bc = 0, pl = 0;
  nbid = 0;
  if (nb < 0) { kind = TASK_BND; bc = -nb - 1; }
  else if ((size_t)nb >= e0 && (size_t)nb < e1) {
    if (!own_left) return false;
    kind = TASK_INT; pl = nb - (int)e0;
  } else { kind = TASK_EXT; nbid = nb; }
  a = TASK_PACK(d - e0, lf, own_left, code, kind, bc, pl);
  f = fid[(size_t)lf * stride + d];
  // faces to other tiles first, then in-tile faces, then boundary faces (as qdg_mesh_upload)
  const int rank = kind == TASK_EXT ? 0 : kind == TASK_INT ? 1 : 2;
  key = (uint32_t)(tile << 4) | (uint32_t)((rank << 2) | lf);
  return true;
}

__global__ void k_task_keys(size_t ne, int stride, int tile_rows, const int* __restrict__ nbr,
                            const int* __restrict__ finfo, const int* __restrict__ fid,
                            uint32_t* __restrict__ key, uint32_t* __restrict__ val, int* __restrict__ count)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * ne) return;
  int a, nbid, f; uint32_t k;
  const bool ok = task_of(i >> 2, (int)(i & 3), stride, tile_rows, ne, nbr, finfo, fid, a, nbid, f, k);
  key[i] = ok ? k : 0xffffffffu; val[i] = (uint32_t)i;
  if (ok) atomicAdd(count, 1);
}

__global__ void k_tile_off(size_t ntask, int ntile, const uint32_t* __restrict__ skey, int* __restrict__ tile_off)
{
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > ntask) return;
  if (r == ntask) { tile_off[ntile] = (int)ntask; return; }
  const int t = (int)(skey[r] >> 4);
  if (r == 0 || (int)(skey[r - 1] >> 4) != t) tile_off[t] = (int)r;
}

__global__ void k_task_fill(size_t ntask, size_t ne, int stride, int tile_rows, int task_stride,
                            const int* __restrict__ nbr, const int* __restrict__ finfo, const int* __restrict__ fid,
                            const uint32_t* __restrict__ sval, const int* __restrict__ tile_off,
                            int* __restrict__ task_a, int* __restrict__ task_nb, int* __restrict__ task_f)
{
  const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ntask) return;
  const size_t i = sval[r];
  int a, nbid, f; uint32_t k;
  task_of(i >> 2, (int)(i & 3), stride, tile_rows, ne, nbr, finfo, fid, a, nbid, f, k);
  const size_t tile = k >> 4;
  const size_t pos = task_stride ? tile * (size_t)task_stride + (r - (size_t)tile_off[tile]) : r;
  task_a[pos] = a; task_nb[pos] = nbid; task_f[pos] = f;
}

// Padded task lists without a global sort: one workgroup per tile.  Within a tile the order is (kind rank,
// local face), then device row -- the key of the sort path (k_task_keys + stable radix sort); a row has at
// most one task per (rank, local face), so a task's position is the number of tasks in lower bins plus the
// number of lower rows in its own bin: 12 wave ballots per wave, popcounts, no sort.  The kernel writes all
// task_stride slots of its tile (unused: -1 / 0 / 0) and the tile's task count.
__global__ __launch_bounds__(256) void k_tile_tasks(size_t ne, int stride, int tile_rows, int task_stride,
                                                    const int* __restrict__ nbr, const int* __restrict__ finfo,
                                                    const int* __restrict__ fid, int* __restrict__ task_a,
                                                    int* __restrict__ task_nb, int* __restrict__ task_f,
                                                    int* __restrict__ tile_cnt)
{
  __shared__ unsigned long long mask[12][4];
  __shared__ int base[13];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const size_t tile = blockIdx.x, d = tile * (size_t)tile_rows + tid;
  const bool row = tid < tile_rows && d < ne;
  int a[4], nbid[4], f[4], bin[4];
#pragma unroll
  for (int lf = 0; lf < 4; ++lf) {
    uint32_t k = 0;
    bin[lf] = -1;
    if (row && task_of(d, lf, stride, tile_rows, ne, nbr, finfo, fid, a[lf], nbid[lf], f[lf], k)) bin[lf] = (int)(k & 15);
  }
#pragma unroll
  for (int b = 0; b < 12; ++b) {
    const int lf = b & 3;
    const unsigned long long m = __ballot(bin[lf] == b);
    if (lane == 0) mask[b][wave] = m;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int b = 0; b < 12; ++b) {
      base[b] = acc;
      acc += __popcll(mask[b][0]) + __popcll(mask[b][1]) + __popcll(mask[b][2]) + __popcll(mask[b][3]);
    }
    base[12] = acc;
    tile_cnt[tile] = acc;
  }
  __syncthreads();
  const size_t slot0 = tile * (size_t)task_stride;
#pragma unroll
  for (int lf = 0; lf < 4; ++lf) {
    const int b = bin[lf];
    if (b < 0) continue;
    int pos = base[b];
    for (int w = 0; w < wave; ++w) pos += __popcll(mask[b][w]);
    pos += __popcll(mask[b][wave] & ((1ull << lane) - 1ull));
    task_a[slot0 + pos] = a[lf]; task_nb[slot0 + pos] = nbid[lf]; task_f[slot0 + pos] = f[lf];
  }
  for (int p = base[12] + tid; p < task_stride; p += 256) {
    task_a[slot0 + p] = -1; task_nb[slot0 + p] = 0; task_f[slot0 + p] = 0;
  }
}

__global__ void k_fill_i32(int* p, size_t n, int v)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void k_iota_i32(int* p, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int)i;
}
__global__ void k_fill_f64(double* p, size_t n, double v)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// sort (key, value) pairs of 32-bit keys; results in ko / vo
static int sort32(uint32_t* ki, uint32_t* ko, uint32_t* vi, uint32_t* vo, size_t n, hipStream_t s)
{
  size_t bytes = 0;
  DHIP(rocprim::radix_sort_pairs(nullptr, bytes, ki, ko, vi, vo, n, 0, 32, s));
  Buf<char> tmp;
  DHIP(tmp.alloc(bytes));
  DHIP(rocprim::radix_sort_pairs(tmp.p, bytes, ki, ko, vi, vo, n, 0, 32, s));
  DHIP(hipStreamSynchronize(s));
  return 0;
}

}  // namespace

// QDG_UPLOAD_STATS=1: wall time of the sections of the device build, on stderr
struct Lap {
  bool on; hipStream_t s; std::chrono::steady_clock::time_point t;
  explicit Lap(hipStream_t st) : on(std::getenv("QDG_UPLOAD_STATS") != nullptr), s(st), t(std::chrono::steady_clock::now()) {}
  void operator()(const char* what)
  {
    if (!on) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "qdg device build: %-40s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};

// BC type per boundary face: bndSurfInt over the configured side sets of each type
// (src/PDE/Integrate/Boundary.cpp:84-90); faces of unconfigured sets get no flux
static int bc_of_faces(qdg_ctx* ctx, const DevFD& fd, std::vector<int>& bcface)
{
  bcface.assign(fd.nbfac, 0);
  for (size_t f = 0; f < fd.nbfac; ++f) {
    int type = 0;
    for (size_t i = 0; i < ctx->bc_sideset.size(); ++i)
      if (ctx->bc_sideset[i] == fd.fset[f]) {
        if (type != 0 && type != ctx->bc_type[i])
          return fail("qdg_mesh_from_connectivity: a side set is configured with two different BC types");
        type = ctx->bc_type[i];
      }
    bcface[f] = type;
  }
  return 0;
}

static int keep_connectivity(qdg_mesh* m, DevFD& fd);

// fd: resident FaceData of the chunk; bcface[nbfac]: BC type of every boundary face (0 = none)
static int dev_build_layout(qdg_ctx* ctx, DevFD& fd, const std::vector<int>& bcface, qdg_mesh** out)
{
  DHIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  Lap lap(s);
  // nie owned tets, ghosts behind them (nie == ne: no ghosts)
  const size_t ne = fd.nelem, nie = fd.nie, nnode = fd.nnode, nf = fd.nipfac, n4 = 4 * ne;
  const size_t stride = (ne + 63) / 64 * 64;
  if (ne > (size_t)(INT32_MAX - 64) / 4 || nf > (size_t)INT32_MAX) return fail("qdg_mesh_from_connectivity: chunk too large");
  std::unique_ptr<qdg_mesh> m(new qdg_mesh);
  m->ctx = ctx;
  m->ndof = ctx->cfg.ndof;
  const int ncomp = ctx->cfg.pde == QDG_PDE_TRANSPORT ? (ctx->cfg.ncomp > 0 ? ctx->cfg.ncomp : 1) : NCOMP;
  m->nprop = ncomp * m->ndof;
  m->nie = nie; m->ne = ne; m->stride = stride;

  Buf<int> d_err, d_count;
  DHIP(d_err.alloc(1)); DHIP(d_count.alloc(3));
  DHIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));
  DHIP(hipMemsetAsync(d_count.p, 0, 3 * sizeof(int), s));

  // ---- device order: Morton curve of the centroids (ties by tet id) --------------------
  Buf<unsigned long long> d_mm;
  DHIP(d_mm.alloc(6));
  {
    const unsigned long long init[6] = { ~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull };
    DHIP(hipMemcpyAsync(d_mm.p, init, sizeof init, hipMemcpyHostToDevice, s));
  }
  k_bbox<<<std::min(nblk(nie), 1024u), 256, 0, s>>>(fd.geoElem.p, nie, d_mm.p);
  unsigned long long hmm[6];
  DHIP(hipMemcpyAsync(hmm, d_mm.p, sizeof hmm, hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  double lo[3], hi[3];
  for (int d = 0; d < 3; ++d) { lo[d] = dord_inv(hmm[d]); hi[d] = dord_inv(hmm[3 + d]); }
  const double ext = std::max({ hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-300 });
  Buf<uint64_t> mkey, mkey2;
  Buf<uint32_t> mval, mval2;
  DHIP(mkey.alloc(ne)); DHIP(mkey2.alloc(ne)); DHIP(mval.alloc(ne)); DHIP(mval2.alloc(ne));
  k_morton<<<nblk(nie), 256, 0, s>>>(fd.geoElem.p, nie, fd.esuel.p, lo[0], lo[1], lo[2], ext, mkey.p, mval.p,
                                    d_count.p + 2, (ne > nie && ctx->opt.halo_depth == 2) ? 1 : 0);
  {
    size_t bytes = 0;
    DHIP(rocprim::radix_sort_pairs(nullptr, bytes, mkey.p, mkey2.p, mval.p, mval2.p, nie, 0, 64, s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    DHIP(rocprim::radix_sort_pairs(tmp.p, bytes, mkey.p, mkey2.p, mval.p, mval2.p, nie, 0, 64, s));
    if (ne > nie) k_ghost_ids<<<nblk(ne - nie), 256, 0, s>>>(mval2.p, nie, ne);   // ghosts keep their rows
    DHIP(hipStreamSynchronize(s));
  }
  lap("Morton order");
  Buf<int> h2d;
  DHIP(h2d.alloc(ne));
  HIPCHK(m->d2h.alloc(ne));
  k_invert<<<nblk(ne), 256, 0, s>>>(mval2.p, ne, h2d.p, m->d2h.p);

  // ---- reference face id per (tet, local face); first touch of nodes and faces ---------
  Buf<int> rface;
  DHIP(rface.alloc(n4));
  k_fill_i32<<<nblk(n4), 256, 0, s>>>(rface.p, n4, -1);
  k_rface<<<nblk(nf), 256, 0, s>>>(nf, fd.esuf.p, fd.esuel.p, fd.inpoel.p, fd.inpofa.p, rface.p, d_err.p);
  // New ids in the order of first touch.  A slot 4 * d + k is the first touch of at most one node and at
  // most one face, so an id is the number of first-touch slots before its own: two prefix sums over the
  // slots instead of sorting (first-touch slot, id) pairs (round 3: 210 ms of a 1.25 s build at 80.9 M tets).
  Buf<uint32_t> fkn, fkf;
  DHIP(fkn.alloc(nnode)); DHIP(fkf.alloc(nf));
  k_fill_u32<<<nblk(nnode), 256, 0, s>>>(fkn.p, nnode, 0xffffffffu);
  k_fill_u32<<<nblk(nf), 256, 0, s>>>(fkf.p, nf, 0xffffffffu);
  {
    int herr = 0;
    DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    if (herr) return fail("qdg_mesh_from_connectivity: inconsistent FaceData (esuf / esuel / inpofa)");
  }
  k_first_touch<<<nblk(n4), 256, 0, s>>>(ne, m->d2h.p, fd.inpoel.p, rface.p, fkn.p, fkf.p);
  Buf<int> flagn, flagf, posn, posf, nnew, fmap;
  DHIP(flagn.alloc(n4)); DHIP(flagf.alloc(n4)); DHIP(posn.alloc(n4)); DHIP(posf.alloc(n4));
  DHIP(nnew.alloc(nnode)); DHIP(fmap.alloc(nf));
  k_fill_i32<<<nblk(nnode), 256, 0, s>>>(nnew.p, nnode, -1);
  k_fill_i32<<<nblk(nf), 256, 0, s>>>(fmap.p, nf, -1);
  k_touch_flags<<<nblk(n4), 256, 0, s>>>(ne, m->d2h.p, fd.inpoel.p, rface.p, fkn.p, fkf.p, flagn.p, flagf.p);
  {
    size_t bytes = 0;
    DHIP(rocprim::exclusive_scan(nullptr, bytes, flagn.p, posn.p, 0, n4, rocprim::plus<int>(), s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    DHIP(rocprim::exclusive_scan(tmp.p, bytes, flagn.p, posn.p, 0, n4, rocprim::plus<int>(), s));
    DHIP(rocprim::exclusive_scan(tmp.p, bytes, flagf.p, posf.p, 0, n4, rocprim::plus<int>(), s));
    DHIP(hipStreamSynchronize(s));
  }
  lap("rface + first touch + scans");
  k_touch_assign<<<nblk(n4), 256, 0, s>>>(ne, m->d2h.p, fd.inpoel.p, rface.p, flagn.p, flagf.p, posn.p, posf.p,
                                         nnew.p, fmap.p, d_count.p);
  int hcount[3];
  DHIP(hipMemcpyAsync(hcount, d_count.p, sizeof hcount, hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  const int ncount = hcount[0], nfd = hcount[1];
  const size_t ninner = (size_t)hcount[2];

  lap("numbering");
  // ---- rows, nodes, faces in device numbering ---------------------------------------------
  Buf<int> d_bc;
  DHIP(d_bc.alloc(std::max<size_t>(bcface.size(), 1)));
  if (!bcface.empty()) DHIP(hipMemcpyAsync(d_bc.p, bcface.data(), bcface.size() * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(m->inpoel.alloc(4 * stride)); HIPCHK(m->nbr.alloc(4 * stride)); HIPCHK(m->finfo.alloc(4 * stride));
  HIPCHK(m->fid.alloc(4 * stride)); HIPCHK(m->vol.alloc(stride));
  k_fill_i32<<<nblk(4 * stride), 256, 0, s>>>(m->inpoel.p, 4 * stride, 0);
  k_fill_i32<<<nblk(4 * stride), 256, 0, s>>>(m->nbr.p, 4 * stride, -1);
  k_fill_i32<<<nblk(4 * stride), 256, 0, s>>>(m->finfo.p, 4 * stride, 0);
  k_fill_i32<<<nblk(4 * stride), 256, 0, s>>>(m->fid.p, 4 * stride, 0);
  k_fill_f64<<<nblk(stride), 256, 0, s>>>(m->vol.p, stride, 1.0);       // (padding rows: volume 1)
  k_layout_rows<<<nblk(nie), 256, 0, s>>>(nie, (int)stride, m->d2h.p, h2d.p, fd.inpoel.p, fd.esuel.p, fd.esuf.p,
                                         rface.p, nnew.p, fmap.p, d_bc.p, fd.geoElem.p, m->inpoel.p, m->nbr.p,
                                         m->finfo.p, m->fid.p, m->vol.p, d_err.p);
  if (ne > nie) k_layout_ghost_nbr<<<nblk(ne - nie), 256, 0, s>>>(nie, ne, (int)stride, h2d.p, fd.esuel.p, m->nbr.p);
  m->ghost_nbr = true;
  const size_t nn1 = (size_t)std::max(ncount, 1), nf1 = (size_t)std::max(nfd, 1);
  HIPCHK(m->x.alloc(nn1)); HIPCHK(m->y.alloc(nn1)); HIPCHK(m->z.alloc(nn1)); HIPCHK(m->xyz4.alloc(4 * nn1));
  HIPCHK(m->farea.alloc(nf1)); HIPCHK(m->fnx.alloc(nf1)); HIPCHK(m->fny.alloc(nf1)); HIPCHK(m->fnz.alloc(nf1));
  HIPCHK(m->fgeo.alloc(4 * nf1));
  k_layout_nodes<<<nblk(nnode), 256, 0, s>>>(nnode, nnew.p, fd.x.p, fd.y.p, fd.z.p, m->x.p, m->y.p, m->z.p, m->xyz4.p);
  k_layout_faces<<<nblk(nf), 256, 0, s>>>(nf, fmap.p, fd.geoFace.p, m->farea.p, m->fnx.p, m->fny.p, m->fnz.p, m->fgeo.p);

  lap("rows / nodes / faces");
  // ---- face tasks of the tile kernels -------------------------------------------------------
  const int tile_rows = TILE;
  const int ntile = (int)((nie + TILE - 1) / TILE);
  const size_t nt4 = 4 * nie;
  const int task_stride = !ctx->cfg.pref ? 4 * TILE_BS : 0;
  HIPCHK(m->tile_off.alloc(ntile + 1)); HIPCHK(m->tile_row.alloc(ntile + 1));
  size_t nslot = 0;
  if (task_stride) {
    // padded lists (every run but the p-adaptive ones): built tile by tile, no sort over 4 * nie keys
    nslot = (size_t)ntile * task_stride;
    HIPCHK(m->task_a.alloc(std::max<size_t>(nslot, 1))); HIPCHK(m->task_nb.alloc(std::max<size_t>(nslot, 1)));
    HIPCHK(m->task_f.alloc(std::max<size_t>(nslot, 1)));
    Buf<int> cnt;
    DHIP(cnt.alloc(ntile + 1));
    DHIP(hipMemsetAsync(cnt.p, 0, (ntile + 1) * sizeof(int), s));
    k_tile_tasks<<<ntile, 256, 0, s>>>(nie, (int)stride, tile_rows, task_stride, m->nbr.p, m->finfo.p, m->fid.p,
                                       m->task_a.p, m->task_nb.p, m->task_f.p, cnt.p);
    size_t bytes = 0;
    DHIP(rocprim::exclusive_scan(nullptr, bytes, cnt.p, m->tile_off.p, 0, (size_t)ntile + 1, rocprim::plus<int>(), s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    DHIP(rocprim::exclusive_scan(tmp.p, bytes, cnt.p, m->tile_off.p, 0, (size_t)ntile + 1, rocprim::plus<int>(), s));
    int herr = 0;
    DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    if (herr == 6) return fail("qdg_mesh_from_connectivity: a free face of an owned tet is in no side set");
    if (herr) return fail("qdg_mesh_from_connectivity: neighbour does not share the face nodes (bad connectivity)");
  } else {
    Buf<uint32_t> tk, tk2, tv, tv2;
    DHIP(tk.alloc(nt4)); DHIP(tk2.alloc(nt4)); DHIP(tv.alloc(nt4)); DHIP(tv2.alloc(nt4));
    Buf<int> d_nt;
    DHIP(d_nt.alloc(1));
    DHIP(hipMemsetAsync(d_nt.p, 0, sizeof(int), s));
    k_task_keys<<<nblk(nt4), 256, 0, s>>>(nie, (int)stride, tile_rows, m->nbr.p, m->finfo.p, m->fid.p, tk.p, tv.p, d_nt.p);
    if (int rc = sort32(tk.p, tk2.p, tv.p, tv2.p, nt4, s)) return rc;
    int ntask = 0, herr = 0;
    DHIP(hipMemcpyAsync(&ntask, d_nt.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    if (herr == 6) return fail("qdg_mesh_from_connectivity: a free face of an owned tet is in no side set");
    if (herr) return fail("qdg_mesh_from_connectivity: neighbour does not share the face nodes (bad connectivity)");
    nslot = (size_t)ntask;
    HIPCHK(m->task_a.alloc(std::max<size_t>(nslot, 1))); HIPCHK(m->task_nb.alloc(std::max<size_t>(nslot, 1)));
    HIPCHK(m->task_f.alloc(std::max<size_t>(nslot, 1)));
    k_tile_off<<<nblk((size_t)ntask + 1), 256, 0, s>>>((size_t)ntask, ntile, tk2.p, m->tile_off.p);
    k_task_fill<<<nblk((size_t)ntask), 256, 0, s>>>((size_t)ntask, nie, (int)stride, tile_rows, task_stride, m->nbr.p,
                                                    m->finfo.p, m->fid.p, tv2.p, m->tile_off.p, m->task_a.p,
                                                    m->task_nb.p, m->task_f.p);
  }
  {
    std::vector<int> rows(ntile + 1);
    for (int t = 0; t <= ntile; ++t) rows[t] = (int)std::min((size_t)t * TILE, nie);
    DHIP(hipMemcpyAsync(m->tile_row.p, rows.data(), rows.size() * sizeof(int), hipMemcpyHostToDevice, s));
    DHIP(hipStreamSynchronize(s));
  }
  DHIP(hipGetLastError());

  lap("face tasks");
  if (int rc = mesh_alloc_state(m.get(), ntile)) return rc;
  m->nnode_used = (size_t)ncount;
  DevMesh& dm = m->dm;
  dm.nie = (int)nie; dm.ne = (int)ne; dm.stride = (int)stride; dm.nnode = ncount; dm.nfac = nfd;
  dm.inpoel = m->inpoel.p; dm.nbr = m->nbr.p; dm.finfo = m->finfo.p; dm.fid = m->fid.p;
  dm.x = m->x.p; dm.y = m->y.p; dm.z = m->z.p;
  dm.farea = m->farea.p; dm.fnx = m->fnx.p; dm.fny = m->fny.p; dm.fnz = m->fnz.p;
  dm.vol = m->vol.p; dm.d2h = m->d2h.p; dm.fgeo = m->fgeo.p; dm.xyz4 = m->xyz4.p;
  int ntile_inner = 0;
  while (ntile_inner < ntile && std::min((size_t)(ntile_inner + 1) * TILE, nie) <= ninner) ++ntile_inner;
  dm.ntile = ntile; dm.ntile_inner = ntile_inner; dm.tile_row = m->tile_row.p; dm.task_stride = task_stride;
  dm.tile_rows = TILE;
  dm.tile_off = m->tile_off.p; dm.task_a = m->task_a.p; dm.task_nb = m->task_nb.p; dm.task_f = m->task_f.p;
  dm.tgeo = nullptr;
  if (task_stride > 0 && ctx->cfg.ndof == 4) {
    HIPCHK(m->tgeo.alloc(4 * nslot));
    launch_task_geo(nslot, m->task_a.p, m->task_f.p, m->fgeo.p, m->tgeo.p, s);
    dm.tgeo = m->tgeo.p;
  }
  dm.blk0 = 0; dm.ninner = (int)ninner; dm.ncomp = ncomp; dm.pde = ctx->cfg.pde; dm.ndofel = nullptr;
  dm.nlim = (int)nie;
  if (ctx->cfg.pref) {
    HIPCHK(m->ndofel.alloc(ne)); HIPCHK(m->ndofel2.alloc(ne));
    k_fill_i32<<<nblk(ne), 256, 0, s>>>(m->ndofel.p, ne, 4);
    dm.ndofel = m->ndofel.p;
  }
  DHIP(hipStreamSynchronize(s));
  lap("state allocation");
  // (a chunk with ghosts is kept only with its tets' global ids: its re-mesh orders the new ghosts by them)
  if (ctx->opt.keep_connectivity && (fd.nie == fd.nelem || fd.gid.p))
    if (int rc = keep_connectivity(m.get(), fd)) return rc;
  *out = m.release();
  return 0;
}

// ======================================================================================
// Uniform 1:8 refinement on the device: same result, array for array, as qdg_refine_uniform
// (qdg_partition.cpp: children order of src/Inciter/AMR/refinement.hpp, midpoints numbered after
// the old nodes in the order the tets meet their edges), computed with one radix sort of the
// 6 * nelem edge keys and two scans, then copied back into a qdg_refined handle.  What a re-mesh
// of ~10 M tets spends on the host otherwise (0.3 s) takes a few tens of ms here.
namespace {
__constant__ int c_edg[6][2] = { {0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3} };   // AB AC AD BC BD CD

__global__ void k_edge_keys(const uint64_t* __restrict__ inpoel, size_t ns, size_t nnode, uint64_t* __restrict__ key,
                            uint32_t* __restrict__ slot, int* __restrict__ err)
{
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  const size_t e = s / 6; const int k = (int)(s - 6 * e);
  const uint64_t a = inpoel[4 * e + c_edg[k][0]], b = inpoel[4 * e + c_edg[k][1]];
  if (a >= nnode || b >= nnode) { *err = 1; key[s] = 0; slot[s] = (uint32_t)s; return; }
  if (a == b) { *err = 2; key[s] = 0; slot[s] = (uint32_t)s; return; }
  key[s] = ((a < b ? a : b) << 32) | (a < b ? b : a);
  slot[s] = (uint32_t)s;
}
// position of the head of every run of equal keys (0 elsewhere; an inclusive max-scan spreads it)
__global__ void k_run_heads(const uint64_t* __restrict__ skey, size_t ns, int* __restrict__ headpos)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  headpos[i] = (i == 0 || skey[i] != skey[i - 1]) ? (int)i : 0;
}
// first[slot] = the smallest slot with the same edge (the sort is stable: the head of the run);
// ishead[slot] = 1 where the slot itself is that one
__global__ void k_first_slot(const uint32_t* __restrict__ sslot, const int* __restrict__ headpos, size_t ns,
                             uint32_t* __restrict__ first, int* __restrict__ ishead)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ns) return;
  const uint32_t s = sslot[i], f = sslot[headpos[i]];
  first[s] = f;
  ishead[s] = (f == s) ? 1 : 0;
}
__global__ void k_midpoints(const uint64_t* __restrict__ key, const uint32_t* __restrict__ first,
                            const int* __restrict__ ishead, const int* __restrict__ rank, size_t ns, size_t nnode,
                            const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                            double* __restrict__ x2, double* __restrict__ y2, double* __restrict__ z2)
{
  const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns || !ishead[s]) return;
  const size_t id = nnode + (size_t)rank[s];
  const uint64_t a = key[s] >> 32, b = key[s] & 0xffffffffu;
  x2[id] = 0.5 * (x[a] + x[b]); y2[id] = 0.5 * (y[a] + y[b]); z2[id] = 0.5 * (z[a] + z[b]);
}
__global__ void k_children(const uint64_t* __restrict__ inpoel, const uint32_t* __restrict__ first,
                           const int* __restrict__ rank, size_t nelem, size_t nnode, uint64_t* __restrict__ out,
                           uint64_t* __restrict__ parent)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nelem) return;
  const uint64_t A = inpoel[4 * e], B = inpoel[4 * e + 1], C = inpoel[4 * e + 2], D = inpoel[4 * e + 3];
  uint64_t M[6];
  for (int k = 0; k < 6; ++k) M[k] = nnode + (uint64_t)rank[first[6 * e + k]];
  const uint64_t AB = M[0], AC = M[1], AD = M[2], BC = M[3], BD = M[4], CD = M[5];
  const uint64_t ch[8][4] = { { A, AB, AC, AD }, { B, BC, AB, BD }, { C, AC, BC, CD }, { D, AD, CD, BD },
                              { BC, CD, AC, BD }, { AB, BD, AC, AD }, { AB, BC, AC, BD }, { AC, BD, CD, AD } };
  for (int k = 0; k < 8; ++k) {
    for (int i = 0; i < 4; ++i) out[4 * (8 * e + k) + i] = ch[k][i];
    if (parent) parent[8 * e + k] = e;
  }
}
__device__ bool edge_mid(const uint64_t* skey, const uint32_t* sslot, const uint32_t* first, const int* rank,
                         size_t ns, size_t nnode, uint64_t a, uint64_t b, uint64_t& m)
{
  const uint64_t key = ((a < b ? a : b) << 32) | (a < b ? b : a);
  size_t lo = 0, hi = ns;
  while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (skey[mid] < key) lo = mid + 1; else hi = mid; }
  if (lo >= ns || skey[lo] != key) return false;
  m = nnode + (uint64_t)rank[first[sslot[lo]]];
  return true;
}
__global__ void k_child_tris(const uint64_t* __restrict__ tri, size_t ntri, const uint64_t* __restrict__ skey,
                             const uint32_t* __restrict__ sslot, const uint32_t* __restrict__ first,
                             const int* __restrict__ rank, size_t ns, size_t nnode, uint64_t* __restrict__ out,
                             int* __restrict__ err)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntri) return;
  const uint64_t a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
  uint64_t ab = 0, bc = 0, ac = 0;
  if (a >= nnode || b >= nnode || c >= nnode || !edge_mid(skey, sslot, first, rank, ns, nnode, a, b, ab) ||
      !edge_mid(skey, sslot, first, rank, ns, nnode, b, c, bc) || !edge_mid(skey, sslot, first, rank, ns, nnode, a, c, ac)) {
    *err = 3;
    return;
  }
  const uint64_t ct[4][3] = { { a, ab, ac }, { b, bc, ab }, { c, ac, bc }, { ab, bc, ac } };
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 3; ++i) out[3 * (4 * t + k) + i] = ct[k][i];
}
}  // namespace

// the refinement itself, on resident arrays: children (k_children's order), the old nodes followed by the
// edge midpoints, the children of the `ntri` triangles d_tri (4 t + k)
struct RefineOut {
  Buf<uint64_t> inpoel2, par, tri2;
  Buf<double> x2, y2, z2;
  size_t nn = 0;
};
static int dev_refine_core(qdg_ctx* ctx, const uint64_t* d_inp, const double* dx, const double* dy, const double* dz,
                           size_t nelem, size_t nnode, const uint64_t* d_tri, size_t ntri, bool want_parent,
                           RefineOut& o)
{
  const size_t ns = 6 * nelem;
  if (nelem == 0) return fail("qdg_refine_uniform_device: empty mesh");
  if (nnode > (size_t)UINT32_MAX / 2 || ns > (size_t)INT32_MAX) return fail("qdg_refine_uniform_device: mesh too large for 32-bit slots");
  hipStream_t s = ctx_stream(ctx);
  Buf<uint64_t> key, skey;
  Buf<uint32_t> slot, sslot, first;
  Buf<int> headpos, headscan, ishead, rank, d_err;
  DHIP(key.alloc(ns)); DHIP(skey.alloc(ns)); DHIP(slot.alloc(ns)); DHIP(sslot.alloc(ns)); DHIP(first.alloc(ns));
  DHIP(headpos.alloc(ns)); DHIP(headscan.alloc(ns)); DHIP(ishead.alloc(ns)); DHIP(rank.alloc(ns)); DHIP(d_err.alloc(1));
  DHIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));
  k_edge_keys<<<nblk(ns), 256, 0, s>>>(d_inp, ns, nnode, key.p, slot.p, d_err.p);
  {
    unsigned bits = 1;
    while (bits < 32 && ((size_t)1 << bits) < nnode) ++bits;
    size_t bytes = 0;
    DHIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, skey.p, slot.p, sslot.p, ns, 0, 32 + bits, s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    DHIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, skey.p, slot.p, sslot.p, ns, 0, 32 + bits, s));
    DHIP(hipStreamSynchronize(s));
  }
  k_run_heads<<<nblk(ns), 256, 0, s>>>(skey.p, ns, headpos.p);
  {
    size_t bytes = 0;
    DHIP(rocprim::inclusive_scan(nullptr, bytes, headpos.p, headscan.p, ns, rocprim::maximum<int>(), s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    DHIP(rocprim::inclusive_scan(tmp.p, bytes, headpos.p, headscan.p, ns, rocprim::maximum<int>(), s));
    DHIP(hipStreamSynchronize(s));
  }
  k_first_slot<<<nblk(ns), 256, 0, s>>>(sslot.p, headscan.p, ns, first.p, ishead.p);
  {
    size_t bytes = 0;
    DHIP(rocprim::exclusive_scan(nullptr, bytes, ishead.p, rank.p, 0, ns, rocprim::plus<int>(), s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    DHIP(rocprim::exclusive_scan(tmp.p, bytes, ishead.p, rank.p, 0, ns, rocprim::plus<int>(), s));
    DHIP(hipStreamSynchronize(s));
  }
  int last_rank = 0, last_head = 0, herr = 0;
  DHIP(hipMemcpyAsync(&last_rank, rank.p + (ns - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(&last_head, ishead.p + (ns - 1), sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  if (herr == 1) return fail("qdg_refine_uniform: inpoel entry out of range");
  if (herr == 2) return fail("qdg_refine_uniform: degenerate tet");
  const size_t nn = nnode + (size_t)last_rank + (size_t)last_head;
  o.nn = nn;
  DHIP(o.x2.alloc(nn)); DHIP(o.y2.alloc(nn)); DHIP(o.z2.alloc(nn));
  DHIP(o.inpoel2.alloc(32 * nelem));
  if (want_parent) DHIP(o.par.alloc(8 * nelem));
  DHIP(hipMemcpyAsync(o.x2.p, dx, nnode * 8, hipMemcpyDeviceToDevice, s));
  DHIP(hipMemcpyAsync(o.y2.p, dy, nnode * 8, hipMemcpyDeviceToDevice, s));
  DHIP(hipMemcpyAsync(o.z2.p, dz, nnode * 8, hipMemcpyDeviceToDevice, s));
  k_midpoints<<<nblk(ns), 256, 0, s>>>(key.p, first.p, ishead.p, rank.p, ns, nnode, dx, dy, dz, o.x2.p, o.y2.p, o.z2.p);
  k_children<<<nblk(nelem), 256, 0, s>>>(d_inp, first.p, rank.p, nelem, nnode, o.inpoel2.p, o.par.p);
  DHIP(o.tri2.alloc(12 * ntri));
  if (ntri)
    k_child_tris<<<nblk(ntri), 256, 0, s>>>(d_tri, ntri, skey.p, sslot.p, first.p, rank.p, ns, nnode, o.tri2.p, d_err.p);
  DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  DHIP(hipGetLastError());
  if (herr == 3) return fail("qdg_refine_uniform: a side-set triangle is not a face of the mesh");
  return 0;
}

extern "C" int qdg_refine_uniform_device(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                                         const double* x, const double* y, const double* z, size_t ntri,
                                         const size_t* tri, qdg_refined** out)
{
  QDG_TRY
  if (!ctx || !inpoel || !x || !y || !z || !out || (ntri && !tri)) return fail("qdg_refine_uniform_device: null argument");
  *out = nullptr;
  if (nelem == 0) return fail("qdg_refine_uniform_device: empty mesh");
  DHIP(hipSetDevice(ctx_device(ctx)));
  hipStream_t s = ctx_stream(ctx);
  qdg::StreamScope scope(s);
  Buf<uint64_t> d_inp, d_tri;
  Buf<double> dx, dy, dz;
  DHIP(d_inp.alloc(4 * nelem)); DHIP(dx.alloc(nnode)); DHIP(dy.alloc(nnode)); DHIP(dz.alloc(nnode));
  DHIP(d_tri.alloc(3 * ntri));
  DHIP(hipMemcpyAsync(d_inp.p, inpoel, 4 * nelem * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(dx.p, x, nnode * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(dy.p, y, nnode * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(dz.p, z, nnode * 8, hipMemcpyHostToDevice, s));
  if (ntri) DHIP(hipMemcpyAsync(d_tri.p, tri, 3 * ntri * 8, hipMemcpyHostToDevice, s));
  RefineOut o;
  if (int rc = dev_refine_core(ctx, d_inp.p, dx.p, dy.p, dz.p, nelem, nnode, d_tri.p, ntri, true, o)) return rc;
  const size_t nn = o.nn;
  std::unique_ptr<qdg_refined> r(new qdg_refined);
  r->nnode = nn;
  r->x.resize(nn); r->y.resize(nn); r->z.resize(nn);
  r->inpoel.resize(32 * nelem); r->parent.resize(8 * nelem); r->tri.resize(12 * ntri);
  if (ntri) DHIP(hipMemcpyAsync(r->tri.data(), o.tri2.p, 12 * ntri * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(r->inpoel.data(), o.inpoel2.p, 32 * nelem * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(r->parent.data(), o.par.p, 8 * nelem * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(r->x.data(), o.x2.p, nn * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(r->y.data(), o.y2.p, nn * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(r->z.data(), o.z2.p, nn * 8, hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  *out = r.release();
  return 0;
  QDG_CATCH
}

// ======================================================================================
// Config 5's re-mesh WITHOUT the host: uniform 1:8 refinement of a resident chunk (no ghosts) from the
// connectivity the mesh handle kept on the device (context option keep_connectivity), the refined mesh's
// FaceData derived from the parent's -- esuel by template: the 8 faces inside a parent join siblings, the 4
// child faces on a parent's face join the children of the parent's neighbour across it that carry the same
// three nodes (src/Inciter/AMR/refinement.hpp:425-536 gives the children; no sort over the 4 * 8 * nelem
// faces) -- boundary faces from the children of the parent's boundary faces, then the same layout build as
// every other mesh, the state handed over child <- parent (src/Inciter/DG.cpp:1597-1605) on the device,
// and the refined mesh copied to the host for its book-keeping by a second host thread on a second stream,
// off the critical path.
struct qdg_mesh::Keep {
  Buf<uint64_t> inpoel, tri;        // caller's numbering; tri = the boundary faces in their order
  Buf<double> x, y, z;
  Buf<int> esuel;
  Buf<uint64_t> gid;                // global tet ids (chunks with ghosts)
  bool orient = false;
  std::vector<int32_t> fset;        // side set of every boundary face
  size_t nelem = 0, nie = 0, nnode = 0, nbfac = 0;
  // halo plan of a chunk with ghosts (qdg_halo_setup): neighbour ranks, ghosts received from each
  std::vector<int32_t> nbr_rank;
  std::vector<size_t> recv_counts;
  std::shared_ptr<qdg_host_copy> pending;   // a host copy still reading these buffers
  ~Keep() { if (pending) pending->join(); }
};

namespace {
void keep_free_fn(qdg_mesh::Keep* k) { delete k; }
struct KeepOn {
  qdg_ctx* c; int saved;
  explicit KeepOn(qdg_ctx* ctx) : c(ctx), saved(ctx->opt.keep_connectivity) { c->opt.keep_connectivity = 1; }
  ~KeepOn() { c->opt.keep_connectivity = saved; }
};

// tables of the 1:8 template (k_children's child order), filled once on the host by enumeration over
// the ten node labels A B C D AB AC AD BC BD CD
struct ChildTables { signed char sib[8][4], pface[8][4], onface[4][4][2]; };
__constant__ ChildTables c_child;

const ChildTables& child_tables()
{
  static const ChildTables t = [] {
    enum { A, B, C, D, AB, AC, AD, BC, BD, CD };
    const int ch[8][4] = { { A, AB, AC, AD }, { B, BC, AB, BD }, { C, AC, BC, CD }, { D, AD, CD, BD },
                           { BC, CD, AC, BD }, { AB, BD, AC, AD }, { AB, BC, AC, BD }, { AC, BD, CD, AD } };
    const int verts[10] = { 1, 2, 4, 8, 1 | 2, 1 | 4, 1 | 8, 2 | 4, 2 | 8, 4 | 8 };   // parent vertices a label touches
    auto fkey = [&](int k, int f) {
      int m = 0;
      for (int j = 0; j < 3; ++j) m |= 1 << ch[k][qdg::LPOFA[f][j]];
      return m;
    };
    ChildTables t{};
    int cnt[4] = { 0, 0, 0, 0 };
    for (int k = 0; k < 8; ++k)
      for (int f = 0; f < 4; ++f) {
        t.sib[k][f] = -1; t.pface[k][f] = -1;
        for (int k2 = 0; k2 < 8; ++k2)
          for (int f2 = 0; f2 < 4; ++f2)
            if (k2 != k && fkey(k2, f2) == fkey(k, f)) t.sib[k][f] = (signed char)k2;
        if (t.sib[k][f] >= 0) continue;
        // on the parent's face lf (opposite its vertex lf) iff no label of the face touches vertex lf
        for (int lf = 0; lf < 4; ++lf) {
          bool on = true;
          for (int j = 0; j < 3; ++j) if (verts[ch[k][qdg::LPOFA[f][j]]] & (1 << lf)) on = false;
          if (on) t.pface[k][f] = (signed char)lf;
        }
        const int lf = t.pface[k][f];
        t.onface[lf][cnt[lf]][0] = (signed char)k; t.onface[lf][cnt[lf]][1] = (signed char)f;
        ++cnt[lf];
      }
    return t;
  }();
  return t;
}

__global__ void k_child_esuel(const uint64_t* __restrict__ inpoel2, const int* __restrict__ esuel_p, size_t n4,
                              int* __restrict__ esuel2, int* __restrict__ err)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const size_t c = i >> 2, e = c >> 3; const int f = (int)(i & 3), k = (int)(c & 7);
  const int sib = c_child.sib[k][f];
  if (sib >= 0) { esuel2[i] = (int)(8 * e + sib); return; }
  const int lf = c_child.pface[k][f];
  const int nb = esuel_p[4 * e + lf];
  if (nb < 0) { esuel2[i] = -1; return; }
  int lf2 = -1;
  for (int q = 0; q < 4; ++q) if (esuel_p[4 * (size_t)nb + q] == (int)e) lf2 = q;
  uint64_t a = inpoel2[4 * c + c_lpofa[f][0]], b = inpoel2[4 * c + c_lpofa[f][1]], d = inpoel2[4 * c + c_lpofa[f][2]], t;
  if (a > b) { t = a; a = b; b = t; }
  if (b > d) { t = b; b = d; d = t; }
  if (a > b) { t = a; a = b; b = t; }
  int found = -1;
  if (lf2 >= 0)
    for (int j = 0; j < 4; ++j) {
      const int k2 = c_child.onface[lf2][j][0], f2 = c_child.onface[lf2][j][1];
      const size_t c2 = 8 * (size_t)nb + k2;
      uint64_t p = inpoel2[4 * c2 + c_lpofa[f2][0]], q = inpoel2[4 * c2 + c_lpofa[f2][1]], r = inpoel2[4 * c2 + c_lpofa[f2][2]];
      if (p > q) { t = p; p = q; q = t; }
      if (q > r) { t = q; q = r; r = t; }
      if (p > q) { t = p; p = q; q = t; }
      if (p == a && q == b && r == d) found = (int)c2;
    }
  if (found < 0) { *err = 1; found = -1; }
  esuel2[i] = found;
}

// sorted keys (a <= b <= c) of n triangles + identity permutation
__global__ void k_tri_keys(const uint64_t* __restrict__ tri, size_t n, uint32_t* __restrict__ a, uint32_t* __restrict__ b,
                           uint32_t* __restrict__ c, uint32_t* __restrict__ perm)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t k0 = (uint32_t)tri[3 * i], k1 = (uint32_t)tri[3 * i + 1], k2 = (uint32_t)tri[3 * i + 2], t;
  if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
  if (k1 > k2) { t = k1; k1 = k2; k2 = t; }
  if (k0 > k1) { t = k0; k0 = k1; k1 = t; }
  a[i] = k0; b[i] = k1; c[i] = k2; perm[i] = (uint32_t)i;
}
__global__ void k_rank_of_child(const uint32_t* __restrict__ parent_rank, const uint32_t* __restrict__ sperm, size_t n,
                                uint32_t* __restrict__ out)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = parent_rank[sperm[i] >> 2];          // child triangle 4 t + k of boundary face t
}
__global__ void k_invert_perm(const int* __restrict__ d2h, size_t n, int* __restrict__ h2d)
{
  const size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d < n) h2d[d2h[d]] = (int)d;
}
}  // namespace

namespace qdg {
void launch_state_transfer(int nrow, int nprop, const int* d2h_to, const int* parent, const int* h2d_from,
                           const double* Ufrom, double* Uto, hipStream_t s);
}

namespace qdg {
void keep_set_plan(qdg_mesh* m, size_t nnbr, const int32_t* nbr_rank, const size_t* recv_off)
{
  if (!m || !m->keep) return;
  m->keep->nbr_rank.assign(nbr_rank, nbr_rank + nnbr);
  m->keep->recv_counts.resize(nnbr);
  for (size_t i = 0; i < nnbr; ++i) m->keep->recv_counts[i] = recv_off[i + 1] - recv_off[i];
}
}  // namespace qdg

// after a build: the chunk's connectivity stays resident with the mesh handle
static int keep_connectivity(qdg_mesh* m, DevFD& fd)
{
  std::unique_ptr<qdg_mesh::Keep> k(new qdg_mesh::Keep);
  k->inpoel.take(fd.inpoel); k->tri.take(fd.tri); k->x.take(fd.x); k->y.take(fd.y); k->z.take(fd.z);
  k->esuel.take(fd.esuel);
  k->gid.take(fd.gid); k->orient = fd.orient;
  k->fset = fd.fset;
  k->nelem = fd.nelem; k->nie = fd.nie; k->nnode = fd.nnode; k->nbfac = fd.nbfac;
  m->keep = k.release();
  m->keep_free = keep_free_fn;
  return 0;
}

extern "C" int qdg_mesh_refine_uniform(qdg_mesh* mesh, qdg_mesh** out, qdg_refined** host_copy)
{
  QDG_TRY
  if (!mesh || !out) return fail("qdg_mesh_refine_uniform: null argument");
  *out = nullptr;
  if (host_copy) *host_copy = nullptr;
  qdg_ctx* ctx = mesh->ctx;
  if (!mesh->keep)
    return fail("qdg_mesh_refine_uniform: the mesh keeps no connectivity on the device (set the context option "
                "keep_connectivity = 1 before building it with qdg_mesh_from_connectivity)");
  if (mesh->ne != mesh->nie) return fail("qdg_mesh_refine_uniform: chunks with ghosts re-mesh through qdg_refine_chunk + qdg_mesh_from_chunk");
  if (mesh->dm.ndofel)
    return fail("qdg_mesh_refine_uniform: p-adaptive runs are not combined with mesh refinement "
                "(DG::resizePostAMR does not carry m_ndof over either)");
  qdg_mesh::Keep& kp = *mesh->keep;
  if (kp.pending) kp.pending->join();
  const size_t nelem = kp.nelem, nnode = kp.nnode, nb = kp.nbfac;
  if (8 * nelem > (size_t)(INT32_MAX - 64) / 4) return fail("qdg_mesh_refine_uniform: refined chunk too large for 32-bit ids");
  DHIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  qdg::StreamScope scope(s);
  Lap lap(s);
  RefineOut o;
  if (int rc = dev_refine_core(ctx, kp.inpoel.p, kp.x.p, kp.y.p, kp.z.p, nelem, nnode, kp.tri.p, nb, false, o)) return rc;
  lap("refinement (edge sort, midpoints, children)");
  DevFD fd;
  fd.nelem = fd.nie = 8 * nelem; fd.nnode = o.nn;
  fd.inpoel.take(o.inpoel2); fd.x.take(o.x2); fd.y.take(o.y2); fd.z.take(o.z2);
  const size_t n4 = 4 * fd.nelem;
  // ---- esuel of the children from the parents' ----
  {
    const ChildTables& t = child_tables();
    DHIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_child), &t, sizeof t, 0, hipMemcpyHostToDevice, s));
    Buf<int> d_err;
    DHIP(d_err.alloc(1));
    DHIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));
    DHIP(fd.esuel.alloc(n4));
    k_child_esuel<<<nblk(n4), 256, 0, s>>>(fd.inpoel.p, kp.esuel.p, n4, fd.esuel.p, d_err.p);
    int herr = 0;
    DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    if (herr) return fail("qdg_mesh_refine_uniform: a child face finds no partner across its parent's face (inconsistent esuel)");
  }
  lap("esuel of the children (template)");
  // ---- boundary faces: the children of the parent's boundary faces are the side-set triangles ----
  const size_t ntri2 = 4 * nb;
  std::vector<int32_t> sets(kp.fset);
  std::sort(sets.begin(), sets.end());
  sets.erase(std::unique(sets.begin(), sets.end()), sets.end());
  if (ntri2 == 0) {
    DHIP(fd.tri.alloc(1)); DHIP(fd.belem.alloc(1));
    fd.nbfac = 0; fd.belem_known = true;
  } else {
    std::vector<uint32_t> hrank(nb);
    for (size_t b = 0; b < nb; ++b) hrank[b] = (uint32_t)(std::lower_bound(sets.begin(), sets.end(), kp.fset[b]) - sets.begin());
    Buf<uint32_t> prank, ka, kb, kc, key, key2, perm, perm2, ta, tb, tc, tr;
    DHIP(prank.alloc(nb)); DHIP(ka.alloc(ntri2)); DHIP(kb.alloc(ntri2)); DHIP(kc.alloc(ntri2)); DHIP(key.alloc(ntri2));
    DHIP(key2.alloc(ntri2)); DHIP(perm.alloc(ntri2)); DHIP(perm2.alloc(ntri2));
    DHIP(ta.alloc(ntri2)); DHIP(tb.alloc(ntri2)); DHIP(tc.alloc(ntri2)); DHIP(tr.alloc(ntri2));
    DHIP(hipMemcpyAsync(prank.p, hrank.data(), nb * 4, hipMemcpyHostToDevice, s));
    k_tri_keys<<<nblk(ntri2), 256, 0, s>>>(o.tri2.p, ntri2, ka.p, kb.p, kc.p, perm.p);
    unsigned bits = 1;
    while (bits < 32 && ((size_t)1 << bits) < fd.nnode) ++bits;
    size_t bytes = 0;
    DHIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key2.p, perm.p, perm2.p, ntri2, 0, bits, s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    const uint32_t* pass[3] = { kc.p, kb.p, ka.p };
    uint32_t *pin = perm.p, *pout = perm2.p;
    for (int ps = 0; ps < 3; ++ps) {
      k_gather<<<nblk(ntri2), 256, 0, s>>>(pass[ps], pin, ntri2, key.p);
      DHIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, key2.p, pin, pout, ntri2, 0, bits, s));
      std::swap(pin, pout);
    }
    k_gather<<<nblk(ntri2), 256, 0, s>>>(ka.p, pin, ntri2, ta.p);
    k_gather<<<nblk(ntri2), 256, 0, s>>>(kb.p, pin, ntri2, tb.p);
    k_gather<<<nblk(ntri2), 256, 0, s>>>(kc.p, pin, ntri2, tc.p);
    k_rank_of_child<<<nblk(ntri2), 256, 0, s>>>(prank.p, pin, ntri2, tr.p);
    if (int rc = dev_bnd_faces_core(ctx, fd, ntri2, ta.p, tb.p, tc.p, tr.p, sets)) return rc;
    if (fd.nbfac != ntri2) return fail("qdg_mesh_refine_uniform: the refined boundary faces do not match the parents'");
  }
  lap("boundary faces of the children");
  if (int rc = dev_faces_geometry(ctx, fd, nullptr)) return rc;
  if (fd.nonpos_vol) return fail("qdg_mesh_refine_uniform: non-positive child volume");
  lap("faces + geometry (device)");
  std::vector<int> bcface;
  if (int rc = bc_of_faces(ctx, fd, bcface)) return rc;
  // side sets of the host copy's triangles (children of the parent's boundary faces, 4 b + k)
  std::unique_ptr<qdg_refined> r;
  if (host_copy) {
    r.reset(new qdg_refined);
    r->nnode = o.nn;
    r->tri_set.resize(ntri2);
    for (size_t b = 0; b < nb; ++b) for (int k = 0; k < 4; ++k) r->tri_set[4 * b + k] = kp.fset[b];
  }
  qdg_mesh* nm = nullptr;
  // (the layout build consumes fd's FaceData; its connectivity moves into the new handle's Keep afterwards)
  int rc = 0;
  {
    KeepOn keep_on(ctx);          // a re-meshed handle always keeps its connectivity (restored on every path)
    rc = dev_build_layout(ctx, fd, bcface, &nm);
  }
  if (rc) return rc;
  std::unique_ptr<qdg_mesh, int (*)(qdg_mesh*)> guard(nm, qdg_mesh_destroy);
  // ---- state: child <- parent, through the two device numberings ----
  {
    Buf<int> h2d_from;
    DHIP(h2d_from.alloc(mesh->ne));
    k_invert_perm<<<nblk(mesh->ne), 256, 0, s>>>(mesh->d2h.p, mesh->ne, h2d_from.p);
    launch_state_transfer((int)nm->ne, nm->nprop, nm->d2h.p, nullptr, h2d_from.p, mesh->Ucur, nm->Ucur, s);
    DHIP(hipGetLastError());
    DHIP(hipStreamSynchronize(s));
    nm->Unp = nullptr; nm->Upending = nullptr;
  }
  lap("state transfer (child <- parent)");
  // ---- host copy of the refined mesh, by a second thread on its own stream ----
  if (host_copy) {
    qdg_mesh::Keep* nk = nm->keep;
    auto hc = std::make_shared<qdg_host_copy>();
    qdg_refined* rp = r.get();
    // the child triangles o.tri2 are not part of the new Keep (its tri = the regenerated boundary faces):
    // the thread owns them
    auto tri2 = std::make_shared<Buf<uint64_t>>();
    tri2->take(o.tri2);
    const int dev = ctx->device;
    const size_t ne2 = fd.nelem, nn2 = o.nn;
    hc->th = std::thread([hc_raw = hc.get(), rp, nk, tri2, dev, ne2, nn2, ntri2] {
      auto chk = [&](hipError_t e) { if (e != hipSuccess && hc_raw->error.empty()) hc_raw->error = hipGetErrorString(e); };
      chk(hipSetDevice(dev));
      hipStream_t s2 = nullptr;
      chk(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      try {                                       // (nothing may leave a thread's function: the host arrays can fail to allocate)
        rp->inpoel.resize(4 * ne2); rp->x.resize(nn2); rp->y.resize(nn2); rp->z.resize(nn2); rp->tri.resize(3 * ntri2);
        chk(hipMemcpyAsync(rp->inpoel.data(), nk->inpoel.p, 4 * ne2 * 8, hipMemcpyDeviceToHost, s2));
        chk(hipMemcpyAsync(rp->x.data(), nk->x.p, nn2 * 8, hipMemcpyDeviceToHost, s2));
        chk(hipMemcpyAsync(rp->y.data(), nk->y.p, nn2 * 8, hipMemcpyDeviceToHost, s2));
        chk(hipMemcpyAsync(rp->z.data(), nk->z.p, nn2 * 8, hipMemcpyDeviceToHost, s2));
        if (ntri2) chk(hipMemcpyAsync(rp->tri.data(), tri2->p, 3 * ntri2 * 8, hipMemcpyDeviceToHost, s2));
        rp->parent.resize(ne2);
        for (size_t c = 0; c < ne2; ++c) rp->parent[c] = c >> 3;
      } catch (const std::exception& ex) {
        if (hc_raw->error.empty()) hc_raw->error = ex.what();
      }
      chk(hipStreamSynchronize(s2));
      if (s2) chk(hipStreamDestroy(s2));
    });
    r->pending = hc;
    nk->pending = hc;
    *host_copy = r.release();
  }
  *out = guard.release();
  return 0;
  QDG_CATCH
}

// ======================================================================================
// Uniform 8:1 derefinement of a resident chunk without ghosts: the inverse of qdg_mesh_refine_uniform for a handle
// whose kept connectivity IS a uniform refinement in this library's order (qdg_derefine_uniform states the rule and
// the reference lines).  Coarse tets from the children's first nodes, structure verified on the device; the coarse
// boundary faces from the refined handle's boundary faces (a child triangle with exactly one old node, the corner of
// its parent triangle, names that triangle through the end points of its two midpoints; same side set); then the
// general build.  The reference's DG has no solution transfer for removed tets (src/Inciter/DG.cpp:1597-1605 handles
// added tets only), so the state of a coarse tet is, by `policy`,
//   QDG_DEREF_FIRST_CHILD  the row of its first child -- the exact inverse of the reference's row copy child <- parent
//                          (a refinement followed by this derefinement returns every DOF);
//   QDG_DEREF_MEAN         the volume-weighted mean of its children's means, higher-order DOFs zero (conservative).
namespace {
__global__ void k_derefine(const uint64_t* __restrict__ ip, size_t np, size_t nnode, uint64_t* __restrict__ out,
                           unsigned long long* __restrict__ mm, int* __restrict__ err)
{
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  const uint64_t* c = ip + 32 * p;
  const uint64_t A = c[0], B = c[4], C = c[8], D = c[12];
  const uint64_t AB = c[1], AC = c[2], AD = c[3], BC = c[5], BD = c[7], CD = c[11];
  const uint64_t want[8][4] = { { A, AB, AC, AD }, { B, BC, AB, BD }, { C, AC, BC, CD }, { D, AD, CD, BD },
                                { BC, CD, AC, BD }, { AB, BD, AC, AD }, { AB, BC, AC, BD }, { AC, BD, CD, AD } };
  bool ok = true;
  for (int k = 0; k < 8; ++k)
    for (int i = 0; i < 4; ++i) ok = ok && c[4 * k + i] == want[k][i] && c[4 * k + i] < nnode;
  if (!ok) { *err = 1; return; }
  out[4 * p] = A; out[4 * p + 1] = B; out[4 * p + 2] = C; out[4 * p + 3] = D;
  unsigned long long hi = A > B ? A : B; hi = C > hi ? C : hi; hi = D > hi ? D : hi;
  unsigned long long lo = AB < AC ? AB : AC; lo = AD < lo ? AD : lo; lo = BC < lo ? BC : lo; lo = BD < lo ? BD : lo; lo = CD < lo ? CD : lo;
  atomicMax(mm, hi); atomicMin(mm + 1, lo);
}
// end points of every midpoint node (all parents of an edge write the same pair)
__global__ void k_midpoint_ends(const uint64_t* __restrict__ ip, size_t np, size_t ncoarse, uint32_t* __restrict__ ends)
{
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  const uint64_t* c = ip + 32 * p;
  const uint64_t V[4] = { c[0], c[4], c[8], c[12] };
  const uint64_t M[6] = { c[1], c[2], c[3], c[5], c[7], c[11] };
  const int E[6][2] = { {0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3} };
  for (int k = 0; k < 6; ++k) {
    ends[2 * (M[k] - ncoarse)] = (uint32_t)V[E[k][0]];
    ends[2 * (M[k] - ncoarse) + 1] = (uint32_t)V[E[k][1]];
  }
}
// a refined boundary face with exactly one old node -> its parent triangle (else ~0)
__global__ void k_parent_tris(const uint64_t* __restrict__ tri, size_t nb, size_t ncoarse, const uint32_t* __restrict__ ends,
                              uint64_t* __restrict__ out)
{
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nb) return;
  const uint64_t q[3] = { tri[3 * b], tri[3 * b + 1], tri[3 * b + 2] };
  int nc = 0, ic = 0;
  for (int i = 0; i < 3; ++i) if (q[i] < ncoarse) { ++nc; ic = i; }
  out[3 * b] = out[3 * b + 1] = out[3 * b + 2] = ~0ull;
  if (nc != 1) return;
  const uint64_t a = q[ic], m1 = q[(ic + 1) % 3], m2 = q[(ic + 2) % 3];
  const uint32_t *e1 = ends + 2 * (m1 - ncoarse), *e2 = ends + 2 * (m2 - ncoarse);
  out[3 * b] = a;
  out[3 * b + 1] = e1[0] == a ? e1[1] : e1[0];
  out[3 * b + 2] = e2[0] == a ? e2[1] : e2[0];
}
__global__ void k_derefine_state(size_t np, int nprop, int ndof, int policy, const int* __restrict__ d2h_c,
                                 const int* __restrict__ h2d_f, const double* __restrict__ vol_f,
                                 const double* __restrict__ Uf, double* __restrict__ Uc)
{
  const size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= np) return;
  const size_t p = (size_t)d2h_c[d];
  if (policy == 0) {
    const size_t r = (size_t)h2d_f[8 * p];
    for (int i = 0; i < nprop; ++i) Uc[d * nprop + i] = Uf[r * nprop + i];
    return;
  }
  double vs = 0.0;
  for (int k = 0; k < 8; ++k) vs += vol_f[h2d_f[8 * p + k]];
  for (int i = 0; i < nprop; ++i) {
    double v = 0.0;
    if (i % ndof == 0) {
      for (int k = 0; k < 8; ++k) { const size_t r = (size_t)h2d_f[8 * p + k]; v += vol_f[r] * Uf[r * nprop + i]; }
      v /= vs;
    }
    Uc[d * nprop + i] = v;
  }
}
}  // namespace

extern "C" int qdg_mesh_derefine_uniform(qdg_mesh* mesh, int policy, qdg_mesh** out)
{
  QDG_TRY
  if (!mesh || !out) return fail("qdg_mesh_derefine_uniform: null argument");
  *out = nullptr;
  if (policy != QDG_DEREF_FIRST_CHILD && policy != QDG_DEREF_MEAN) return fail("qdg_mesh_derefine_uniform: unknown policy");
  qdg_ctx* ctx = mesh->ctx;
  if (!mesh->keep) return fail("qdg_mesh_derefine_uniform: the mesh keeps no connectivity on the device (context option keep_connectivity = 1)");
  if (mesh->ne != mesh->nie) return fail("qdg_mesh_derefine_uniform: chunks with ghosts are not derefined here");
  if (mesh->dm.ndofel) return fail("qdg_mesh_derefine_uniform: p-adaptive runs are not combined with mesh refinement");
  qdg_mesh::Keep& kp = *mesh->keep;
  if (kp.pending) kp.pending->join();
  const size_t nelem = kp.nelem, nnode = kp.nnode, nb = kp.nbfac;
  if (nelem == 0 || nelem % 8 != 0) return fail("qdg_mesh_derefine_uniform: the number of tets is not a multiple of 8");
  const size_t np = nelem / 8;
  DHIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  qdg::StreamScope scope(s);
  if (int rc = mesh_flush_carry(mesh)) return rc;
  DevFD fd;
  fd.nelem = fd.nie = np;
  Buf<unsigned long long> d_mm;
  Buf<int> d_err;
  DHIP(fd.inpoel.alloc(4 * np)); DHIP(d_mm.alloc(2)); DHIP(d_err.alloc(1));
  {
    const unsigned long long init[2] = { 0ull, ~0ull };
    DHIP(hipMemcpyAsync(d_mm.p, init, sizeof init, hipMemcpyHostToDevice, s));
    DHIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));
  }
  k_derefine<<<nblk(np), 256, 0, s>>>(kp.inpoel.p, np, nnode, fd.inpoel.p, d_mm.p, d_err.p);
  unsigned long long hmm[2]; int herr = 0;
  DHIP(hipMemcpyAsync(hmm, d_mm.p, sizeof hmm, hipMemcpyDeviceToHost, s));
  DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  if (herr) return fail("qdg_mesh_derefine_uniform: the kept connectivity is not a uniform refinement in the order of "
                        "qdg_refine_uniform (eight children per tet, 8 e + k)");
  const size_t ncoarse = (size_t)hmm[0] + 1;
  if (hmm[1] < ncoarse) return fail("qdg_mesh_derefine_uniform: a midpoint node is numbered before a corner node");
  fd.nnode = ncoarse;
  DHIP(fd.x.alloc(ncoarse)); DHIP(fd.y.alloc(ncoarse)); DHIP(fd.z.alloc(ncoarse));
  DHIP(hipMemcpyAsync(fd.x.p, kp.x.p, ncoarse * 8, hipMemcpyDeviceToDevice, s));
  DHIP(hipMemcpyAsync(fd.y.p, kp.y.p, ncoarse * 8, hipMemcpyDeviceToDevice, s));
  DHIP(hipMemcpyAsync(fd.z.p, kp.z.p, ncoarse * 8, hipMemcpyDeviceToDevice, s));
  // ---- coarse side-set triangles from the refined boundary faces (a surface-sized detour through the host) ----
  std::vector<size_t> tri; std::vector<int32_t> tset;
  if (nb) {
    Buf<uint32_t> ends; Buf<uint64_t> ptri;
    DHIP(ends.alloc(2 * (nnode - ncoarse))); DHIP(ptri.alloc(3 * nb));
    k_midpoint_ends<<<nblk(np), 256, 0, s>>>(kp.inpoel.p, np, ncoarse, ends.p);
    k_parent_tris<<<nblk(nb), 256, 0, s>>>(kp.tri.p, nb, ncoarse, ends.p, ptri.p);
    std::vector<uint64_t> h(3 * nb);
    DHIP(hipMemcpyAsync(h.data(), ptri.p, 3 * nb * 8, hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    for (size_t b = 0; b < nb; ++b)
      if (h[3 * b] != ~0ull) { tri.push_back(h[3 * b]); tri.push_back(h[3 * b + 1]); tri.push_back(h[3 * b + 2]); tset.push_back(kp.fset[b]); }
  }
  {
    SortedFaces sf;
    if (int rc = dev_esuel_by_sort(ctx, fd, sf)) return rc;
  }
  if (int rc = dev_bnd_faces(ctx, fd, tset.size(), tri.data(), tset.data())) return rc;
  if (int rc = dev_faces_geometry(ctx, fd, nullptr)) return rc;
  if (fd.nonpos_vol) return fail("qdg_mesh_derefine_uniform: non-positive parent volume");
  std::vector<int> bcface;
  if (int rc = bc_of_faces(ctx, fd, bcface)) return rc;
  qdg_mesh* nm = nullptr;
  int rc = 0;
  {
    KeepOn keep_on(ctx);
    rc = dev_build_layout(ctx, fd, bcface, &nm);
  }
  if (rc) return rc;
  std::unique_ptr<qdg_mesh, int (*)(qdg_mesh*)> guard(nm, qdg_mesh_destroy);
  {
    Buf<int> h2d_from;
    DHIP(h2d_from.alloc(mesh->ne));
    k_invert_perm<<<nblk(mesh->ne), 256, 0, s>>>(mesh->d2h.p, mesh->ne, h2d_from.p);
    k_derefine_state<<<nblk(np), 256, 0, s>>>(np, nm->nprop, nm->ndof, policy, nm->d2h.p, h2d_from.p, mesh->vol.p, mesh->Ucur, nm->Ucur);
    DHIP(hipGetLastError());
    DHIP(hipStreamSynchronize(s));
    nm->Unp = nullptr; nm->Upending = nullptr;
  }
  *out = guard.release();
  return 0;
  QDG_CATCH
}

// ======================================================================================
// Device side of qdg_state_transfer / qdg_state_migrate: the parent (or source) row of every row of `to`
// found on the device -- a conversion kernel for a caller-supplied list, a sort + binary search for the
// match by global tet id -- instead of host loops and hash maps over all tets.
namespace {
int dev_sort_pairs64(uint64_t* ki, uint64_t* ko, uint32_t* vi, uint32_t* vo, size_t n, hipStream_t s);   // (below)
__global__ void k_parent_convert(const size_t* __restrict__ par64, size_t n, size_t nfrom, int* __restrict__ par32,
                                 int* __restrict__ err)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const size_t p = par64[c];
  if (p == (size_t)-1) { par32[c] = -1; return; }            // QDG_NO_ROW: the row is left as it is
  if (p >= nfrom) { *err = 1; par32[c] = -1; return; }
  par32[c] = (int)p;
}
__global__ void k_iota_u32(uint32_t* p, size_t n)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}
__global__ void k_match_gid(const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sval, size_t nfrom,
                            const uint64_t* __restrict__ to_gid, size_t nto, int* __restrict__ par,
                            unsigned long long* __restrict__ count)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool hit = false;
  if (c < nto) {
    const uint64_t g = to_gid[c];
    size_t lo = 0, hi = nfrom;
    while (lo < hi) { const size_t mid = (lo + hi) >> 1; if (skey[mid] < g) lo = mid + 1; else hi = mid; }
    hit = lo < nfrom && skey[lo] == g;
    par[c] = hit ? (int)sval[lo] : -1;
  }
  const unsigned long long b = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (unsigned long long)__popcll(b));
}
}  // namespace

namespace qdg {
// rows [0, nrow) of `to` (device order) <- rows of `from` by the DEVICE parent list d_par (caller's numbering of
// both meshes; -1: row left as it is)
static int transfer_rows(qdg_mesh* from, qdg_mesh* to, size_t nrow, const int* d_par)
{
  hipStream_t s = to->ctx->stream;
  if (int rc = mesh_flush_carry(from)) return rc;
  if (int rc = mesh_flush_carry(to)) return rc;
  Buf<int> h2d_from;
  DHIP(h2d_from.alloc(from->ne));
  k_invert_perm<<<nblk(from->ne), 256, 0, s>>>(from->d2h.p, from->ne, h2d_from.p);
  launch_state_transfer((int)nrow, to->nprop, to->d2h.p, d_par, h2d_from.p, from->Ucur, to->Ucur, s);
  DHIP(hipGetLastError());
  DHIP(hipStreamSynchronize(s));
  to->Unp = nullptr; to->Upending = nullptr;
  to->slab_ready_for = nullptr;        // the send slab no longer holds this state's rows (qdg_step_comm)
  return 0;
}

int dev_state_transfer(qdg_mesh* from, qdg_mesh* to, const size_t* parent_of_child)
{
  qdg_ctx* ctx = to->ctx;
  DHIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  qdg::StreamScope scope(s);
  const size_t n = to->ne;
  Buf<size_t> p64;
  Buf<int> p32, d_err;
  DHIP(p64.alloc(n)); DHIP(p32.alloc(n)); DHIP(d_err.alloc(1));
  DHIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));
  DHIP(hipMemcpyAsync(p64.p, parent_of_child, n * sizeof(size_t), hipMemcpyHostToDevice, s));
  k_parent_convert<<<nblk(n), 256, 0, s>>>(p64.p, n, from->ne, p32.p, d_err.p);
  int herr = 0;
  DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  if (herr) return fail("qdg_state_transfer: parent id out of range");
  return transfer_rows(from, to, n, p32.p);
}

int dev_state_migrate(qdg_mesh* from, const size_t* from_gid, qdg_mesh* to, const size_t* to_gid, size_t* nmoved)
{
  qdg_ctx* ctx = to->ctx;
  DHIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  qdg::StreamScope scope(s);
  const size_t nf = from->nie, nt = to->nie;               // owned rows only
  if (nf > (size_t)UINT32_MAX) return fail("qdg_state_migrate: chunk too large");
  Buf<uint64_t> fk, fk2, tk;
  Buf<uint32_t> fv, fv2;
  Buf<int> par;
  Buf<unsigned long long> cnt;
  DHIP(fk.alloc(nf)); DHIP(fk2.alloc(nf)); DHIP(fv.alloc(nf)); DHIP(fv2.alloc(nf)); DHIP(tk.alloc(nt));
  DHIP(par.alloc(nt)); DHIP(cnt.alloc(1));
  DHIP(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), s));
  DHIP(hipMemcpyAsync(fk.p, from_gid, nf * 8, hipMemcpyHostToDevice, s));
  DHIP(hipMemcpyAsync(tk.p, to_gid, nt * 8, hipMemcpyHostToDevice, s));
  k_iota_u32<<<nblk(nf), 256, 0, s>>>(fv.p, nf);
  if (int rc = dev_sort_pairs64(fk.p, fk2.p, fv.p, fv2.p, nf, s)) return rc;
  k_match_gid<<<nblk(nt), 256, 0, s>>>(fk2.p, fv2.p, nf, tk.p, nt, par.p, cnt.p);
  unsigned long long n = 0;
  DHIP(hipMemcpyAsync(&n, cnt.p, sizeof n, hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  if (nmoved) *nmoved = (size_t)n;
  if (n == 0) return 0;
  return transfer_rows(from, to, nt, par.p);
}
}  // namespace qdg

// ======================================================================================
// The same for ONE RANK's chunk with its ghost layer (config 5 on a decomposition): what qdg_refine_chunk
// derives on the host -- the children of the owned tets, the new ghost layer (children of old ghosts that
// share a face with an owned child, grouped by owner, ordered by global child id 8 * gid(parent) + k), the
// new halo plan (per neighbour the owned children next to its ghosts, ordered by global child id), nodes
// renumbered in ascending order of their ids in the refined chunk -- computed on the device from the
// connectivity, esuel, global ids and plan the handle keeps, without communication: both ranks of a pair
// derive the same sets.  Then the common chunk build, the halo set-up and the state of the owned tets.
namespace {
__global__ void k_ghost_child_flag(const int* __restrict__ esuel2, size_t nown, size_t nall, int* __restrict__ flag)
{
  const size_t c = nown + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nall) return;
  int f = 0;
  for (int q = 0; q < 4; ++q) { const int nb = esuel2[4 * c + q]; f |= (nb >= 0 && (size_t)nb < nown) ? 1 : 0; }
  flag[c - nown] = f;
}
// the sort keys below hold a global CHILD id 8 * gid + k in 48 bits: flag a parent id that does not fit
__global__ void k_check_gid48(const uint64_t* __restrict__ gid, size_t n, int* __restrict__ err)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && gid[i] >= (1ull << 45)) *err = 1;
}
// keys of the new ghosts: (owner index << 48) | global child id; value = child index
__global__ void k_ghost_keys(const int* __restrict__ flag, const int* __restrict__ pos, size_t nown, size_t nall,
                             size_t nie_p, const uint64_t* __restrict__ gid_p, const size_t* __restrict__ recv_off,
                             int nnbr, uint64_t* __restrict__ key, uint32_t* __restrict__ val)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nall - nown || !flag[i]) return;
  const size_t c = nown + i, p = c >> 3, g = p - nie_p;         // g: index of the old ghost
  int ow = 0;
  while (ow + 1 < nnbr && g >= recv_off[ow + 1]) ++ow;
  key[pos[i]] = ((uint64_t)ow << 48) | (8 * gid_p[p] + (c & 7));
  val[pos[i]] = (uint32_t)c;
}
// (owner << 48 | global id of the owned child) for every (new ghost, face to an owned child); ~0 elsewhere
__global__ void k_send_keys(const uint64_t* __restrict__ gkey, const uint32_t* __restrict__ gchild, size_t ng,
                            const int* __restrict__ esuel2, size_t nown, const uint64_t* __restrict__ gid_p,
                            uint64_t* __restrict__ skey, uint32_t* __restrict__ sval)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * ng) return;
  const size_t j = i >> 2; const int q = (int)(i & 3);
  const uint32_t c = gchild[j];
  const int nb = esuel2[4 * (size_t)c + q];
  uint64_t k = ~0ull; uint32_t v = 0;
  if (nb >= 0 && (size_t)nb < nown) {
    k = (gkey[j] & 0xffff000000000000ull) | (8 * gid_p[(size_t)nb >> 3] + ((uint32_t)nb & 7));
    v = (uint32_t)nb;
  }
  skey[i] = k; sval[i] = v;
}
// ---- two ghost layers: the tets within two faces of the owned | ghost interface ("zone") and their children ----
// one hop of the zone: an owned tet joins when a neighbour is in it (in -> out: no race)
__global__ void k_zone_hop(const int* __restrict__ esuel, size_t nie, size_t nunk, const int* __restrict__ in, int* __restrict__ out)
{
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nunk) return;
  int z = e >= nie ? 1 : in[e];
  if (!z)
    for (int f = 0; f < 4; ++f) { const int nb = esuel[4 * e + f]; z |= (nb >= 0 && in[nb]) ? 1 : 0; }
  out[e] = z;
}
// the zone's children as a mesh of their own: adjacency in subset numbering (8 * ppos[parent] + k; -1 outside),
// owner rank (mine: INT32_MIN), global child id, child index in the refined chunk
__global__ void k_zone_children(size_t nunk, size_t nie, const int* __restrict__ zone, const int* __restrict__ ppos,
                                const int* __restrict__ esuel2, const uint64_t* __restrict__ gid_p,
                                const size_t* __restrict__ recv_off, const int32_t* __restrict__ entry_rank, int nentry,
                                int* __restrict__ s_esuel, int32_t* __restrict__ s_owner, uint64_t* __restrict__ s_gid,
                                uint32_t* __restrict__ s_child)
{
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= 8 * nunk) return;
  const size_t p = c >> 3;
  if (!zone[p]) return;
  const size_t i = 8 * (size_t)ppos[p] + (c & 7);
  for (int f = 0; f < 4; ++f) {
    const int d = esuel2[4 * c + f];
    s_esuel[4 * i + f] = (d >= 0 && zone[d >> 3]) ? 8 * ppos[d >> 3] + (d & 7) : -1;
  }
  int32_t ow = INT32_MIN;
  if (p >= nie) {
    const size_t g = p - nie;
    int en = 0;
    while (en + 1 < nentry && g >= recv_off[en + 1]) ++en;
    ow = entry_rank[en];
  }
  s_owner[i] = ow; s_gid[i] = 8 * gid_p[p] + (c & 7); s_child[i] = (uint32_t)c;
}

__global__ void k_mark_unique(const uint64_t* __restrict__ skey, size_t n, int* __restrict__ flag)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (skey[i] != ~0ull && (i == 0 || skey[i] != skey[i - 1])) ? 1 : 0;
}
__global__ void k_compact_pairs(const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sval,
                                const int* __restrict__ flag, const int* __restrict__ pos, size_t n,
                                uint64_t* __restrict__ okey, uint32_t* __restrict__ oval)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !flag[i]) return;
  okey[pos[i]] = skey[i]; oval[pos[i]] = sval[i];
}
// kept tets: owned children [0, nown) in order, then the new ghosts in their sorted order
__global__ void k_kept_pos(const uint32_t* __restrict__ gchild, size_t ng, size_t nown, int* __restrict__ pos_of_child)
{
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < ng) pos_of_child[gchild[j]] = (int)(nown + j);
}
__global__ void k_mark_nodes(const uint64_t* __restrict__ inpoel2, const uint32_t* __restrict__ gchild, size_t nown,
                             size_t nkept, int* __restrict__ used)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * nkept) return;
  const size_t k = i >> 2, c = k < nown ? k : gchild[k - nown];
  used[inpoel2[4 * c + (i & 3)]] = 1;
}
__global__ void k_kept_tets(const uint64_t* __restrict__ inpoel2, const int* __restrict__ esuel2,
                            const uint32_t* __restrict__ gchild, const int* __restrict__ pos_of_child,
                            const int* __restrict__ used, const int* __restrict__ nodepos, size_t nown, size_t nkept,
                            const uint64_t* __restrict__ gid_p, uint64_t* __restrict__ inp_out,
                            int* __restrict__ esuel_out, uint64_t* __restrict__ gid_out, uint64_t* __restrict__ par_out)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 4 * nkept) return;
  const size_t k = i >> 2, c = k < nown ? k : gchild[k - nown];
  const int q = (int)(i & 3);
  inp_out[i] = (uint64_t)nodepos[inpoel2[4 * c + q]];
  const int nb = esuel2[4 * c + q];
  esuel_out[i] = nb < 0 ? -1 : pos_of_child[nb];
  if (q == 0) { gid_out[k] = 8 * gid_p[c >> 3] + (c & 7); par_out[k] = c >> 3; }
}
__global__ void k_compact_nodes(const int* __restrict__ used, const int* __restrict__ nodepos, size_t nn,
                                const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                                double* __restrict__ ox, double* __restrict__ oy, double* __restrict__ oz)
{
  const size_t n = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nn || !used[n]) return;
  ox[nodepos[n]] = x[n]; oy[nodepos[n]] = y[n]; oz[nodepos[n]] = z[n];
}
__global__ void k_remap_u64(uint64_t* __restrict__ a, size_t n, const int* __restrict__ nodepos)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (uint64_t)nodepos[a[i]];
}

static int dev_scan_int(const int* in, int* out, size_t n, hipStream_t s)
{
  size_t bytes = 0;
  DHIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, n, rocprim::plus<int>(), s));
  Buf<char> tmp;
  DHIP(tmp.alloc(bytes));
  DHIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, 0, n, rocprim::plus<int>(), s));
  DHIP(hipStreamSynchronize(s));
  return 0;
}
int dev_sort_pairs64(uint64_t* ki, uint64_t* ko, uint32_t* vi, uint32_t* vo, size_t n, hipStream_t s)
{
  size_t bytes = 0;
  DHIP(rocprim::radix_sort_pairs(nullptr, bytes, ki, ko, vi, vo, n, 0, 64, s));
  Buf<char> tmp;
  DHIP(tmp.alloc(bytes));
  DHIP(rocprim::radix_sort_pairs(tmp.p, bytes, ki, ko, vi, vo, n, 0, 64, s));
  DHIP(hipStreamSynchronize(s));
  return 0;
}
}  // namespace

extern "C" int qdg_mesh_refine_chunk(qdg_mesh* mesh, qdg_mesh** out, qdg_chunk_refined** host_copy, int copy_mesh)
{
  QDG_TRY
  if (!mesh || !out) return fail("qdg_mesh_refine_chunk: null argument");
  *out = nullptr;
  if (host_copy) *host_copy = nullptr;
  qdg_ctx* ctx = mesh->ctx;
  if (!mesh->keep || !mesh->keep->gid.p)
    return fail("qdg_mesh_refine_chunk: the mesh keeps no connectivity / global ids on the device (context option "
                "keep_connectivity = 1, built by qdg_mesh_from_chunk_gid)");
  if (mesh->dm.ndofel) return fail("qdg_mesh_refine_chunk: p-adaptive runs are not combined with mesh refinement");
  qdg_mesh::Keep& kp = *mesh->keep;
  if (kp.pending) kp.pending->join();
  const size_t nunk = kp.nelem, nie = kp.nie, nnode = kp.nnode, nb = kp.nbfac;
  size_t nnbr = kp.nbr_rank.size();
  const bool deep = mesh->nghost1 > 0;            // two ghost layers: the new layers and plan by the shared rule, below
  if (nunk > nie && (nnbr == 0 || mesh->nnbr != nnbr))
    return fail("qdg_mesh_refine_chunk: the chunk has ghosts: call qdg_halo_setup before the re-mesh");
  if (8 * nunk > (size_t)(INT32_MAX - 64) / 4) return fail("qdg_mesh_refine_chunk: refined chunk too large for 32-bit ids");
  if (nnbr >= 65536) return fail("qdg_mesh_refine_chunk: more than 65535 neighbour ranks");   // (sort keys: owner << 48 | child id)
  DHIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  qdg::StreamScope scope(s);
  Lap lap(s);
  // ---- refinement of ALL tets of the chunk (owned and ghosts: the midpoints are numbered over both) ----
  RefineOut o;
  if (int rc = dev_refine_core(ctx, kp.inpoel.p, kp.x.p, kp.y.p, kp.z.p, nunk, nnode, kp.tri.p, nb, false, o)) return rc;
  const size_t nown = 8 * nie, nall = 8 * nunk, n4 = 4 * nall;
  Buf<int> esuel2, d_err;
  DHIP(esuel2.alloc(n4)); DHIP(d_err.alloc(1));
  {
    const ChildTables& t = child_tables();
    DHIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_child), &t, sizeof t, 0, hipMemcpyHostToDevice, s));
    DHIP(hipMemsetAsync(d_err.p, 0, sizeof(int), s));
    k_check_gid48<<<nblk(nunk), 256, 0, s>>>(kp.gid.p, nunk, d_err.p);
    int herr = 0;
    DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    if (herr) return fail("qdg_mesh_refine_chunk: a global tet id is >= 2^45 (the children's ids 8 * gid + k must stay below 2^48)");
    k_child_esuel<<<nblk(n4), 256, 0, s>>>(o.inpoel2.p, kp.esuel.p, n4, esuel2.p, d_err.p);
    DHIP(hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    if (herr) return fail("qdg_mesh_refine_chunk: a child face finds no partner across its parent's face");
  }
  lap("refinement + esuel of the children");
  // ---- the new ghost layer ----
  const size_t ngc = nall - nown;                       // ghost children
  std::vector<size_t> recv_off(nnbr + 1, 0);
  for (size_t i = 0; i < nnbr; ++i) recv_off[i + 1] = recv_off[i] + kp.recv_counts[i];
  if (recv_off[nnbr] != nunk - nie) return fail("qdg_mesh_refine_chunk: the kept halo plan does not match the ghost count");
  Buf<size_t> d_roff;
  DHIP(d_roff.alloc(nnbr + 1));
  DHIP(hipMemcpyAsync(d_roff.p, recv_off.data(), (nnbr + 1) * sizeof(size_t), hipMemcpyHostToDevice, s));
  Buf<int> gflag, gpos;
  Buf<uint64_t> gkey, gkey2;
  Buf<uint32_t> gval, gval2;
  size_t ng = 0;
  if (ngc && !deep) {
    DHIP(gflag.alloc(ngc + 1)); DHIP(gpos.alloc(ngc + 1));
    DHIP(hipMemsetAsync(gflag.p + ngc, 0, sizeof(int), s));
    k_ghost_child_flag<<<nblk(ngc), 256, 0, s>>>(esuel2.p, nown, nall, gflag.p);
    if (int rc = dev_scan_int(gflag.p, gpos.p, ngc + 1, s)) return rc;
    int cnt = 0;
    DHIP(hipMemcpyAsync(&cnt, gpos.p + ngc, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    ng = (size_t)cnt;
  }
  DHIP(gkey.alloc(ng)); DHIP(gkey2.alloc(ng)); DHIP(gval.alloc(ng)); DHIP(gval2.alloc(ng));
  std::vector<size_t> new_recv(nnbr, 0), send_off(nnbr + 1, 0), send_list;
  std::vector<int32_t> new_rank(kp.nbr_rank), new_layer(nnbr, 1);
  size_t new_nghost1 = 0;
  if (deep) {
    // Two ghost layers.  The rule of qdg_chunk_build_depth (qdg_ghost_plan_build) applied to the children of the
    // tets within two faces of the old interface -- nothing farther in can enter a layer or a send list (a child lies
    // at least as many faces from a foreign child as its parent from a foreign tet).  That set is surface-sized: it
    // is cut out on the device, its adjacency / owners / global ids go to the host, and the plan comes back.
    Buf<int> za, zb, ppos;
    DHIP(za.alloc(nunk + 1)); DHIP(zb.alloc(nunk + 1)); DHIP(ppos.alloc(nunk + 1));
    DHIP(hipMemsetAsync(za.p, 0, (nunk + 1) * sizeof(int), s));
    DHIP(hipMemsetAsync(zb.p, 0, (nunk + 1) * sizeof(int), s));
    k_zone_hop<<<nblk(nunk), 256, 0, s>>>(kp.esuel.p, nie, nunk, za.p, zb.p);      // the ghosts
    k_zone_hop<<<nblk(nunk), 256, 0, s>>>(kp.esuel.p, nie, nunk, zb.p, za.p);      // + owned tets next to one
    k_zone_hop<<<nblk(nunk), 256, 0, s>>>(kp.esuel.p, nie, nunk, za.p, zb.p);      // + their owned neighbours
    DHIP(hipMemcpyAsync(za.p, zb.p, nunk * sizeof(int), hipMemcpyDeviceToDevice, s));
    if (int rc = dev_scan_int(za.p, ppos.p, nunk + 1, s)) return rc;
    int nz = 0;
    DHIP(hipMemcpyAsync(&nz, ppos.p + nunk, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    const size_t nsub = 8 * (size_t)nz;
    Buf<int> s_es; Buf<int32_t> s_ow, d_er; Buf<uint64_t> s_gid; Buf<uint32_t> s_ch;
    DHIP(s_es.alloc(4 * nsub)); DHIP(s_ow.alloc(nsub)); DHIP(s_gid.alloc(nsub)); DHIP(s_ch.alloc(nsub)); DHIP(d_er.alloc(nnbr));
    DHIP(hipMemcpyAsync(d_er.p, kp.nbr_rank.data(), nnbr * sizeof(int32_t), hipMemcpyHostToDevice, s));
    k_zone_children<<<nblk(nall), 256, 0, s>>>(nunk, nie, za.p, ppos.p, esuel2.p, kp.gid.p, d_roff.p, d_er.p, (int)nnbr,
                                              s_es.p, s_ow.p, s_gid.p, s_ch.p);
    std::vector<int> h_es(4 * nsub); std::vector<int32_t> h_ow(nsub); std::vector<size_t> h_gid(nsub); std::vector<uint32_t> h_ch(nsub);
    DHIP(hipMemcpyAsync(h_es.data(), s_es.p, 4 * nsub * sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipMemcpyAsync(h_ow.data(), s_ow.p, nsub * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DHIP(hipMemcpyAsync(h_gid.data(), s_gid.p, nsub * 8, hipMemcpyDeviceToHost, s));
    DHIP(hipMemcpyAsync(h_ch.data(), s_ch.p, nsub * 4, hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    qdg_ghost_plan* gp = nullptr;
    if (int rc = qdg_ghost_plan_build(nsub, h_es.data(), h_ow.data(), h_gid.data(), INT32_MIN, 2, &gp)) return rc;
    std::unique_ptr<qdg_ghost_plan, int (*)(qdg_ghost_plan*)> gpg(gp, qdg_ghost_plan_destroy);
    size_t nsend = 0, nent = 0;
    if (int rc = qdg_ghost_plan_sizes(gp, &ng, &new_nghost1, &nent, &nsend)) return rc;
    std::vector<size_t> gh(std::max<size_t>(ng, 1)), roff(nent + 1), se(std::max<size_t>(nsend, 1));
    new_rank.assign(nent, 0); new_layer.assign(nent, 0); send_off.assign(nent + 1, 0);
    if (int rc = qdg_ghost_plan_get(gp, gh.data(), new_rank.data(), new_layer.data(), roff.data(), send_off.data(), se.data())) return rc;
    nnbr = nent;
    new_recv.assign(nent, 0);
    for (size_t i = 0; i < nent; ++i) new_recv[i] = roff[i + 1] - roff[i];
    send_list.resize(nsend);
    for (size_t j = 0; j < nsend; ++j) send_list[j] = h_ch[se[j]];        // an owned child's index is its new local id
    std::vector<uint32_t> gchild(std::max<size_t>(ng, 1));
    for (size_t i = 0; i < ng; ++i) gchild[i] = h_ch[gh[i]];
    DHIP(gval2.alloc(ng));
    if (ng) DHIP(hipMemcpyAsync(gval2.p, gchild.data(), ng * 4, hipMemcpyHostToDevice, s));
    DHIP(hipStreamSynchronize(s));
  } else if (ng) {
    k_ghost_keys<<<nblk(ngc), 256, 0, s>>>(gflag.p, gpos.p, nown, nall, nie, kp.gid.p, d_roff.p, (int)nnbr, gkey.p, gval.p);
    if (int rc = dev_sort_pairs64(gkey.p, gkey2.p, gval.p, gval2.p, ng, s)) return rc;
    std::vector<uint64_t> hk(ng);
    DHIP(hipMemcpyAsync(hk.data(), gkey2.p, ng * 8, hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    for (uint64_t k : hk) ++new_recv[(size_t)(k >> 48)];
    // send lists: owned children next to the new ghosts, per owner by global child id, each once
    const size_t ns4 = 4 * ng;
    Buf<uint64_t> sk, sk2, sk3;
    Buf<uint32_t> sv, sv2, sv3;
    Buf<int> uf, up;
    DHIP(sk.alloc(ns4)); DHIP(sk2.alloc(ns4)); DHIP(sv.alloc(ns4)); DHIP(sv2.alloc(ns4));
    DHIP(uf.alloc(ns4 + 1)); DHIP(up.alloc(ns4 + 1));
    k_send_keys<<<nblk(ns4), 256, 0, s>>>(gkey2.p, gval2.p, ng, esuel2.p, nown, kp.gid.p, sk.p, sv.p);
    if (int rc = dev_sort_pairs64(sk.p, sk2.p, sv.p, sv2.p, ns4, s)) return rc;
    DHIP(hipMemsetAsync(uf.p + ns4, 0, sizeof(int), s));
    k_mark_unique<<<nblk(ns4), 256, 0, s>>>(sk2.p, ns4, uf.p);
    if (int rc = dev_scan_int(uf.p, up.p, ns4 + 1, s)) return rc;
    int nsend = 0;
    DHIP(hipMemcpyAsync(&nsend, up.p + ns4, sizeof(int), hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    DHIP(sk3.alloc((size_t)nsend)); DHIP(sv3.alloc((size_t)nsend));
    k_compact_pairs<<<nblk(ns4), 256, 0, s>>>(sk2.p, sv2.p, uf.p, up.p, ns4, sk3.p, sv3.p);
    std::vector<uint64_t> hsk((size_t)nsend); std::vector<uint32_t> hsv((size_t)nsend);
    DHIP(hipMemcpyAsync(hsk.data(), sk3.p, (size_t)nsend * 8, hipMemcpyDeviceToHost, s));
    DHIP(hipMemcpyAsync(hsv.data(), sv3.p, (size_t)nsend * 4, hipMemcpyDeviceToHost, s));
    DHIP(hipStreamSynchronize(s));
    send_list.resize((size_t)nsend);
    for (size_t i = 0; i < (size_t)nsend; ++i) { ++send_off[(size_t)(hsk[i] >> 48) + 1]; send_list[i] = hsv[i]; }
    for (size_t q = 0; q < nnbr; ++q) send_off[q + 1] += send_off[q];
  }
  lap("new ghost layer and halo plan");
  // ---- the kept tets, nodes renumbered in ascending order of their ids in the refined chunk ----
  const size_t nkept = nown + ng, nn_all = o.nn;
  Buf<int> pos_of_child, used, nodepos;
  DHIP(pos_of_child.alloc(nall)); DHIP(used.alloc(nn_all + 1)); DHIP(nodepos.alloc(nn_all + 1));
  k_fill_i32<<<nblk(nall), 256, 0, s>>>(pos_of_child.p, nall, -1);
  k_iota_i32<<<nblk(nown), 256, 0, s>>>(pos_of_child.p, nown);            // owned children keep their index
  if (ng) k_kept_pos<<<nblk(ng), 256, 0, s>>>(gval2.p, ng, nown, pos_of_child.p);
  DHIP(hipMemsetAsync(used.p, 0, (nn_all + 1) * sizeof(int), s));
  k_mark_nodes<<<nblk(4 * nkept), 256, 0, s>>>(o.inpoel2.p, gval2.p, nown, nkept, used.p);
  if (int rc = dev_scan_int(used.p, nodepos.p, nn_all + 1, s)) return rc;
  int nn2 = 0;
  DHIP(hipMemcpyAsync(&nn2, nodepos.p + nn_all, sizeof(int), hipMemcpyDeviceToHost, s));
  DHIP(hipStreamSynchronize(s));
  DevFD fd;
  fd.nelem = nkept; fd.nie = nown; fd.nnode = (size_t)nn2;
  Buf<uint64_t> par_out;
  DHIP(fd.inpoel.alloc(4 * nkept)); DHIP(fd.esuel.alloc(4 * nkept)); DHIP(fd.gid.alloc(nkept)); DHIP(par_out.alloc(nkept));
  DHIP(fd.x.alloc((size_t)nn2)); DHIP(fd.y.alloc((size_t)nn2)); DHIP(fd.z.alloc((size_t)nn2));
  fd.orient = kp.orient;
  k_kept_tets<<<nblk(4 * nkept), 256, 0, s>>>(o.inpoel2.p, esuel2.p, gval2.p, pos_of_child.p, used.p, nodepos.p, nown, nkept,
                                             kp.gid.p, fd.inpoel.p, fd.esuel.p, fd.gid.p, par_out.p);
  k_compact_nodes<<<nblk(nn_all), 256, 0, s>>>(used.p, nodepos.p, nn_all, o.x2.p, o.y2.p, o.z2.p, fd.x.p, fd.y.p, fd.z.p);
  const size_t ntri2 = 4 * nb;
  if (ntri2) k_remap_u64<<<nblk(3 * ntri2), 256, 0, s>>>(o.tri2.p, 3 * ntri2, nodepos.p);   // (all on owned children: kept)
  lap("kept tets, node renumbering");
  // ---- boundary faces of the owned children ----
  std::vector<int32_t> sets(kp.fset);
  std::sort(sets.begin(), sets.end());
  sets.erase(std::unique(sets.begin(), sets.end()), sets.end());
  if (ntri2 == 0) {
    DHIP(fd.tri.alloc(1)); DHIP(fd.belem.alloc(1));
    fd.nbfac = 0; fd.belem_known = true;
  } else {
    std::vector<uint32_t> hrank(nb);
    for (size_t b = 0; b < nb; ++b) hrank[b] = (uint32_t)(std::lower_bound(sets.begin(), sets.end(), kp.fset[b]) - sets.begin());
    Buf<uint32_t> prank, ka, kb, kc, key, key2, perm, perm2, ta, tb, tc, tr;
    DHIP(prank.alloc(nb)); DHIP(ka.alloc(ntri2)); DHIP(kb.alloc(ntri2)); DHIP(kc.alloc(ntri2)); DHIP(key.alloc(ntri2));
    DHIP(key2.alloc(ntri2)); DHIP(perm.alloc(ntri2)); DHIP(perm2.alloc(ntri2));
    DHIP(ta.alloc(ntri2)); DHIP(tb.alloc(ntri2)); DHIP(tc.alloc(ntri2)); DHIP(tr.alloc(ntri2));
    DHIP(hipMemcpyAsync(prank.p, hrank.data(), nb * 4, hipMemcpyHostToDevice, s));
    k_tri_keys<<<nblk(ntri2), 256, 0, s>>>(o.tri2.p, ntri2, ka.p, kb.p, kc.p, perm.p);
    unsigned bits = 1;
    while (bits < 32 && ((size_t)1 << bits) < fd.nnode) ++bits;
    size_t bytes = 0;
    DHIP(rocprim::radix_sort_pairs(nullptr, bytes, key.p, key2.p, perm.p, perm2.p, ntri2, 0, bits, s));
    Buf<char> tmp;
    DHIP(tmp.alloc(bytes));
    const uint32_t* pass[3] = { kc.p, kb.p, ka.p };
    uint32_t *pin = perm.p, *pout = perm2.p;
    for (int ps = 0; ps < 3; ++ps) {
      k_gather<<<nblk(ntri2), 256, 0, s>>>(pass[ps], pin, ntri2, key.p);
      DHIP(rocprim::radix_sort_pairs(tmp.p, bytes, key.p, key2.p, pin, pout, ntri2, 0, bits, s));
      std::swap(pin, pout);
    }
    k_gather<<<nblk(ntri2), 256, 0, s>>>(ka.p, pin, ntri2, ta.p);
    k_gather<<<nblk(ntri2), 256, 0, s>>>(kb.p, pin, ntri2, tb.p);
    k_gather<<<nblk(ntri2), 256, 0, s>>>(kc.p, pin, ntri2, tc.p);
    k_rank_of_child<<<nblk(ntri2), 256, 0, s>>>(prank.p, pin, ntri2, tr.p);
    if (int rc = dev_bnd_faces_core(ctx, fd, ntri2, ta.p, tb.p, tc.p, tr.p, sets)) return rc;
    if (fd.nbfac != ntri2) return fail("qdg_mesh_refine_chunk: the refined boundary faces do not match the parents'");
  }
  if (int rc = dev_faces_geometry(ctx, fd, nullptr)) return rc;
  if (fd.nonpos_vol) return fail("qdg_mesh_refine_chunk: non-positive child volume");
  lap("boundary faces, faces + geometry");
  std::vector<int> bcface;
  if (int rc = bc_of_faces(ctx, fd, bcface)) return rc;
  // the host's copy of what it needs for its book-keeping, taken before the layout build consumes fd
  std::unique_ptr<qdg_chunk_refined> hc;
  if (host_copy) {
    hc.reset(new qdg_chunk_refined);
    hc->nielem = nown; hc->nunk = nkept; hc->nnode = (size_t)nn2;
    hc->gid.resize(nkept); hc->parent.resize(nkept);
    DHIP(hipMemcpyAsync(hc->gid.data(), fd.gid.p, nkept * 8, hipMemcpyDeviceToHost, s));
    DHIP(hipMemcpyAsync(hc->parent.data(), par_out.p, nkept * 8, hipMemcpyDeviceToHost, s));
    hc->send_off = send_off; hc->send_list = send_list; hc->recv_counts = new_recv;
    hc->nbr_rank = new_rank; hc->nbr_layer = new_layer; hc->nghost1 = deep ? new_nghost1 : ng;
    if (copy_mesh) {
      hc->inpoel.resize(4 * nkept); hc->x.resize((size_t)nn2); hc->y.resize((size_t)nn2); hc->z.resize((size_t)nn2);
      hc->tri.resize(3 * ntri2); hc->tri_set.resize(ntri2);
      DHIP(hipMemcpyAsync(hc->inpoel.data(), fd.inpoel.p, 4 * nkept * 8, hipMemcpyDeviceToHost, s));
      DHIP(hipMemcpyAsync(hc->x.data(), fd.x.p, (size_t)nn2 * 8, hipMemcpyDeviceToHost, s));
      DHIP(hipMemcpyAsync(hc->y.data(), fd.y.p, (size_t)nn2 * 8, hipMemcpyDeviceToHost, s));
      DHIP(hipMemcpyAsync(hc->z.data(), fd.z.p, (size_t)nn2 * 8, hipMemcpyDeviceToHost, s));
      if (ntri2) DHIP(hipMemcpyAsync(hc->tri.data(), o.tri2.p, 3 * ntri2 * 8, hipMemcpyDeviceToHost, s));
      for (size_t b = 0; b < nb; ++b) for (int k = 0; k < 4; ++k) hc->tri_set[4 * b + k] = kp.fset[b];
    }
    DHIP(hipStreamSynchronize(s));
  }
  qdg_mesh* nm = nullptr;
  int rc = 0;
  {
    KeepOn keep_on(ctx);          // a re-meshed handle always keeps its connectivity (restored on every path)
    rc = dev_build_layout(ctx, fd, bcface, &nm);
  }
  if (rc) return rc;
  std::unique_ptr<qdg_mesh, int (*)(qdg_mesh*)> guard(nm, qdg_mesh_destroy);
  lap("layout");
  // ---- halo plan of the new chunk ----
  {
    std::vector<size_t> roff(nnbr + 1, 0);
    for (size_t q = 0; q < nnbr; ++q) roff[q + 1] = roff[q] + new_recv[q];
    if (int rc2 = qdg_halo_setup(nm, nnbr, new_rank.data(), send_off.data(), send_list.data(), roff.data())) return rc2;
    if (deep) if (int rc2 = qdg_halo_set_depth(nm, new_nghost1)) return rc2;
  }
  // ---- state of the owned tets: child <- parent (ghost rows arrive with the next exchange) ----
  {
    Buf<int> h2d_from;
    DHIP(h2d_from.alloc(mesh->ne));
    k_invert_perm<<<nblk(mesh->ne), 256, 0, s>>>(mesh->d2h.p, mesh->ne, h2d_from.p);
    launch_state_transfer((int)nm->nie, nm->nprop, nm->d2h.p, nullptr, h2d_from.p, mesh->Ucur, nm->Ucur, s);
    DHIP(hipGetLastError());
    DHIP(hipStreamSynchronize(s));
    nm->Unp = nullptr; nm->Upending = nullptr;
  }
  lap("halo set-up + state transfer");
  if (host_copy) *host_copy = hc.release();
  *out = guard.release();
  return 0;
  QDG_CATCH
}

extern "C" int qdg_mesh_from_connectivity(qdg_ctx* ctx, size_t nelem, size_t nnode, const size_t* inpoel,
                                          const double* x, const double* y, const double* z,
                                          size_t ntri, const size_t* tri, const int32_t* tri_set,
                                          qdg_mesh** out)
{
  return qdg_mesh_from_chunk(ctx, nelem, nelem, nnode, inpoel, x, y, z, ntri, tri, tri_set, out);
}

extern "C" int qdg_mesh_from_chunk(qdg_ctx* ctx, size_t nielem, size_t nelem, size_t nnode, const size_t* inpoel,
                                   const double* x, const double* y, const double* z,
                                   size_t ntri, const size_t* tri, const int32_t* tri_set, qdg_mesh** out)
{
  return qdg_mesh_from_chunk_gid(ctx, nielem, nelem, nnode, inpoel, x, y, z, ntri, tri, tri_set, nullptr, out);
}

extern "C" int qdg_mesh_from_chunk_gid(qdg_ctx* ctx, size_t nielem, size_t nelem, size_t nnode, const size_t* inpoel,
                                       const double* x, const double* y, const double* z,
                                       size_t ntri, const size_t* tri, const int32_t* tri_set,
                                       const size_t* elem_gid, qdg_mesh** out)
{
  QDG_TRY
  if (!ctx || !out || !inpoel || !x || !y || !z) return fail("qdg_mesh_from_connectivity: null argument");
  if (ntri > 0 && (!tri || !tri_set)) return fail("qdg_mesh_from_connectivity: null side-set arrays");
  if (nielem == 0 || nielem > nelem) return fail("qdg_mesh_from_chunk: need 0 < nielem <= nelem");
  *out = nullptr;
  if (nielem < nelem && ctx->opt.host_layout)
    return fail("qdg_mesh_from_chunk: option host_layout covers chunks without ghosts only");
  if (!ctx->opt.host_layout) {
    // Everything on the GPU: boundary faces regenerated from the side-set triangles, FaceData,
    // geometry and the device layout.  Only the connectivity, the coordinates and the side-set
    // triangles cross PCIe.
    Lap lap(ctx->stream);
    DevFD fd;
    if (int rc = dev_upload_mesh(ctx, nelem, nnode, inpoel, x, y, z, fd)) return rc;
    fd.nie = nielem;
    if (elem_gid) {
      DHIP(fd.gid.alloc(nelem));
      DHIP(hipMemcpyAsync(fd.gid.p, elem_gid, nelem * 8, hipMemcpyHostToDevice, ctx->stream));
      fd.orient = ctx->opt.orient_by_gid != 0;
    }
    lap("validation + upload of inpoel, coord");
    {
      SortedFaces sf;                  // (its 5 x 4 * nelem words are released before the layout is built)
      if (int rc = dev_esuel_by_sort(ctx, fd, sf)) return rc;
    }
    lap("esuel (device: face sort)");
    if (int rc = dev_bnd_faces(ctx, fd, ntri, tri, tri_set)) return rc;
    lap("boundary faces (device)");
    if (int rc = dev_faces_geometry(ctx, fd, nullptr)) return rc;
    if (fd.nonpos_vol)
      return fail("qdg_mesh_from_connectivity: non-positive element volume (inverted or degenerate tet; the "
                  "reference asserts a positive Jacobian, src/Mesh/DerivedData.cpp:1478-1480)");
    lap("faces + geometry (device)");
    std::vector<int> bcface;
    if (int rc = bc_of_faces(ctx, fd, bcface)) return rc;
    return dev_build_layout(ctx, fd, bcface, out);
  }
  // QDG_HOST_LAYOUT=1: boundary faces on the host (qdg_bnd_faces), FaceData on the device, copied
  // back, layout by qdg_mesh_upload (A/B runs, equivalence tests)
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
  return qdg_mesh_upload_gid(ctx, nelem, nelem, nnode, inpoel, x, y, z, nbfac, nipfac, esuf.data(),
                             esuel.data(), inpofa.data(), geoFace.data(), geoElem.data(), &bf, elem_gid, out);
  QDG_CATCH
}
