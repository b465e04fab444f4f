// qdg_meshdata.cpp -- host-side mirror of inciter::FaceData's constructor and
// of the geometry generators, producing arrays IDENTICAL in content and order
// to the reference's (src/Inciter/FaceData.cpp:19-41 and
// src/Mesh/DerivedData.cpp:937-1491), but with O(n log n) sort-based face
// matching instead of the reference's elements-surrounding-points walk, so
// that 10^7..10^8-tet chunks are set up in seconds.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/qdg.h"
#include "qdg_host.hpp"

namespace qdg {

// src/Mesh/DerivedData.hpp:36
const int LPOFA[4][3] = { {1, 2, 3}, {2, 0, 3}, {3, 0, 1}, {0, 2, 1} };

namespace {

struct FaceKey {
  uint64_t a, b, c;   // sorted node ids
  uint64_t ef;        // 4*element + local face
  bool operator<(const FaceKey& o) const
  {
    if (a != o.a) return a < o.a;
    if (b != o.b) return b < o.b;
    if (c != o.c) return c < o.c;
    return ef < o.ef;
  }
  bool same(const FaceKey& o) const { return a == o.a && b == o.b && c == o.c; }
};

inline void sort3(uint64_t& a, uint64_t& b, uint64_t& c)
{
  if (a > b) std::swap(a, b);
  if (b > c) std::swap(b, c);
  if (a > b) std::swap(a, b);
}

}  // namespace

}  // namespace qdg

using namespace qdg;

// elements surrounding elements (src/Mesh/DerivedData.cpp:937-1051):
// esuel[4*e+f] = tet sharing local face LPOFA[f] of e, or -1
extern "C" int qdg_gen_esuel(size_t nelem, const size_t* inpoel, int* esuel)
{
  QDG_TRY
  if (nelem > (size_t)INT32_MAX / 4) return fail("qdg_gen_esuel: too many elements for int ids");
  std::vector<FaceKey> keys(4 * nelem);
  for (size_t e = 0; e < nelem; ++e)
    for (int f = 0; f < 4; ++f) {
      FaceKey& k = keys[4 * e + f];
      k.a = inpoel[4 * e + LPOFA[f][0]];
      k.b = inpoel[4 * e + LPOFA[f][1]];
      k.c = inpoel[4 * e + LPOFA[f][2]];
      sort3(k.a, k.b, k.c);
      k.ef = 4 * e + f;
    }
  std::sort(keys.begin(), keys.end());
  for (size_t i = 0; i < 4 * nelem; ++i) esuel[i] = -1;
  for (size_t i = 0; i + 1 < keys.size(); ++i) {
    if (keys[i].same(keys[i + 1])) {
      if (i + 2 < keys.size() && keys[i].same(keys[i + 2]))
        return fail("qdg_gen_esuel: face shared by more than two tets (non-manifold mesh)");
      esuel[keys[i].ef] = (int)(keys[i + 1].ef / 4);
      esuel[keys[i + 1].ef] = (int)(keys[i].ef / 4);
      ++i;
    }
  }
  return 0;
  QDG_CATCH
}

// src/Mesh/DerivedData.cpp:1053-1093
extern "C" size_t qdg_gen_nipfac(size_t nelem, size_t nbfac, const int* esuel)
{
  size_t n = 0;
  for (size_t e = 0; e < nelem; ++e)
    for (int f = 0; f < 4; ++f) {
      const int j = esuel[4 * e + f];
      if (j != -1 && e < (size_t)j) ++n;
    }
  return n + nbfac;
}

// src/Mesh/DerivedData.cpp:1153-1218
extern "C" int qdg_gen_inpofa(size_t nelem, size_t nbfac, const size_t* inpoel,
                              const size_t* triinpoel, const int* esuel, size_t* inpofa)
{
  QDG_TRY
  size_t ic = 3 * nbfac;
  for (size_t e = 0; e < nelem; ++e)
    for (int f = 0; f < 4; ++f) {
      const int j = esuel[4 * e + f];
      if (j != -1 && e < (size_t)j) {
        inpofa[ic] = inpoel[4 * e + LPOFA[f][0]];
        inpofa[ic + 1] = inpoel[4 * e + LPOFA[f][1]];
        inpofa[ic + 2] = inpoel[4 * e + LPOFA[f][2]];
        ic += 3;
      }
    }
  for (size_t i = 0; i < 3 * nbfac; ++i) inpofa[i] = triinpoel[i];
  return 0;
  QDG_CATCH
}

// host element of every boundary face (src/Mesh/DerivedData.cpp:1220-1290):
// the tet that contains all three face nodes
extern "C" int qdg_gen_belem(size_t nelem, size_t nbfac, const size_t* inpoel,
                             const size_t* inpofa, size_t* belem)
{
  QDG_TRY
  std::vector<FaceKey> keys;
  keys.reserve(4 * nelem + nbfac);
  for (size_t e = 0; e < nelem; ++e)
    for (int f = 0; f < 4; ++f) {
      FaceKey k;
      k.a = inpoel[4 * e + LPOFA[f][0]];
      k.b = inpoel[4 * e + LPOFA[f][1]];
      k.c = inpoel[4 * e + LPOFA[f][2]];
      sort3(k.a, k.b, k.c);
      k.ef = 4 * e + f;
      keys.push_back(k);
    }
  std::sort(keys.begin(), keys.end());
  for (size_t f = 0; f < nbfac; ++f) {
    FaceKey k;
    k.a = inpofa[3 * f]; k.b = inpofa[3 * f + 1]; k.c = inpofa[3 * f + 2];
    sort3(k.a, k.b, k.c);
    k.ef = 0;
    auto it = std::lower_bound(keys.begin(), keys.end(), k);
    if (it == keys.end() || !it->same(k))
      return fail("qdg_gen_belem: boundary face " + std::to_string(f) + " is not a face of any tet");
    belem[f] = it->ef / 4;
  }
  return 0;
  QDG_CATCH
}

// src/Mesh/DerivedData.cpp:1095-1151
extern "C" int qdg_gen_esuf(size_t nelem, size_t nbfac, const size_t* belem, const int* esuel,
                            int* esuf)
{
  QDG_TRY
  size_t ic = 2 * nbfac;
  for (size_t e = 0; e < nelem; ++e)
    for (int f = 0; f < 4; ++f) {
      const int j = esuel[4 * e + f];
      if (j != -1 && e < (size_t)j) {
        esuf[ic] = (int)e;
        esuf[ic + 1] = j;
        ic += 2;
      }
    }
  for (size_t f = 0; f < nbfac; ++f) {
    esuf[2 * f] = (int)belem[f];
    esuf[2 * f + 1] = -1;
  }
  return 0;
  QDG_CATCH
}

// src/Mesh/DerivedData.cpp:1292-1434: area (Heron), unit normal, centroid
extern "C" int qdg_gen_geoface(size_t nfac, const size_t* inpofa, const double* x,
                               const double* y, const double* z, double* geoFace)
{
  QDG_TRY
  for (size_t f = 0; f < nfac; ++f) {
    const size_t a = inpofa[3 * f], b = inpofa[3 * f + 1], c = inpofa[3 * f + 2];
    const double X[3] = { x[a], x[b], x[c] }, Y[3] = { y[a], y[b], y[c] }, Z[3] = { z[a], z[b], z[c] };
    const double sa = std::sqrt((X[1]-X[0])*(X[1]-X[0]) + (Y[1]-Y[0])*(Y[1]-Y[0]) + (Z[1]-Z[0])*(Z[1]-Z[0]));
    const double sb = std::sqrt((X[2]-X[1])*(X[2]-X[1]) + (Y[2]-Y[1])*(Y[2]-Y[1]) + (Z[2]-Z[1])*(Z[2]-Z[1]));
    const double sc = std::sqrt((X[0]-X[2])*(X[0]-X[2]) + (Y[0]-Y[2])*(Y[0]-Y[2]) + (Z[0]-Z[2])*(Z[0]-Z[2]));
    const double sp = 0.5 * (sa + sb + sc);
    const double ax = X[1]-X[0], ay = Y[1]-Y[0], az = Z[1]-Z[0];
    const double bx = X[2]-X[0], by = Y[2]-Y[0], bz = Z[2]-Z[0];
    const double nx = ay * bz - az * by, ny = -(ax * bz - az * bx), nz = ax * by - ay * bx;
    const double fa = std::sqrt(nx * nx + ny * ny + nz * nz);
    double* g = geoFace + 7 * f;
    g[0] = std::sqrt(sp * (sp - sa) * (sp - sb) * (sp - sc));
    g[1] = nx / fa; g[2] = ny / fa; g[3] = nz / fa;
    g[4] = (X[0] + X[1] + X[2]) / 3.0;
    g[5] = (Y[0] + Y[1] + Y[2]) / 3.0;
    g[6] = (Z[0] + Z[1] + Z[2]) / 3.0;
  }
  return 0;
  QDG_CATCH
}

// src/Mesh/DerivedData.cpp:1436-1491: volume triple(ba,ca,da)/6 and centroid
extern "C" int qdg_gen_geoelem(size_t nelem, const size_t* inpoel, const double* x,
                               const double* y, const double* z, double* geoElem)
{
  QDG_TRY
  for (size_t e = 0; e < nelem; ++e) {
    const size_t A = inpoel[4*e], B = inpoel[4*e+1], C = inpoel[4*e+2], D = inpoel[4*e+3];
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
  return 0;
  QDG_CATCH
}

// Boundary-face regeneration of the mesh loader
// (src/Inciter/Partitioner.cpp:357-393): side-set triangles are only
// order-independent keys; every tet, in order, contributes its faces
// {0,2,1},{0,1,3},{0,3,2},{1,2,3} that match a key, in that node order.
// Faces come back grouped by ascending side set id (the std::map order of
// FaceData::m_bface), within a set in tet order.
extern "C" int qdg_bnd_faces(size_t nelem, const size_t* inpoel, size_t ntri, const size_t* tri,
                             const int32_t* tri_set, size_t* nbfac, size_t* triinpoel,
                             int32_t* face_set)
{
  QDG_TRY
  static const int F[4][3] = { {0, 2, 1}, {0, 1, 3}, {0, 3, 2}, {1, 2, 3} };
  std::vector<FaceKey> keys(ntri);
  for (size_t i = 0; i < ntri; ++i) {
    keys[i].a = tri[3 * i]; keys[i].b = tri[3 * i + 1]; keys[i].c = tri[3 * i + 2];
    sort3(keys[i].a, keys[i].b, keys[i].c);
    keys[i].ef = (uint64_t)(int64_t)tri_set[i];
  }
  std::sort(keys.begin(), keys.end());
  struct Hit { int32_t set; uint64_t order; size_t n[3]; };
  std::vector<Hit> hits;
  hits.reserve(ntri);
  for (size_t e = 0; e < nelem; ++e)
    for (int f = 0; f < 4; ++f) {
      FaceKey k;
      const size_t n0 = inpoel[4 * e + F[f][0]], n1 = inpoel[4 * e + F[f][1]], n2 = inpoel[4 * e + F[f][2]];
      k.a = n0; k.b = n1; k.c = n2; k.ef = 0;
      sort3(k.a, k.b, k.c);
      // first key with the same node triple (any set id)
      auto it = std::lower_bound(keys.begin(), keys.end(), k,
        [](const FaceKey& p, const FaceKey& q) {
          if (p.a != q.a) return p.a < q.a;
          if (p.b != q.b) return p.b < q.b;
          return p.c < q.c; });
      if (it != keys.end() && it->same(k)) {
        // a triangle listed in several side sets keeps the LAST insertion of
        // the reference's faceside hash map; ascending set order => largest id
        auto last = it;
        while (last + 1 != keys.end() && (last + 1)->same(k)) ++last;
        hits.push_back({ (int32_t)(int64_t)last->ef, (uint64_t)(4 * e + f), { n0, n1, n2 } });
      }
    }
  std::stable_sort(hits.begin(), hits.end(),
                   [](const Hit& p, const Hit& q) { return p.set < q.set; });
  if (hits.size() > ntri) return fail("qdg_bnd_faces: more matching tet faces than side-set triangles");
  *nbfac = hits.size();
  for (size_t i = 0; i < hits.size(); ++i) {
    triinpoel[3 * i] = hits[i].n[0];
    triinpoel[3 * i + 1] = hits[i].n[1];
    triinpoel[3 * i + 2] = hits[i].n[2];
    face_set[i] = hits[i].set;
  }
  return 0;
  QDG_CATCH
}
