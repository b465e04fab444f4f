// qdg_exo.cpp -- element-field output in ExodusII layout on the netCDF classic format with
// 64-bit offsets (CDF-2), so that a run's fields can be compared with the reference's golden
// files by its own regression harness (exodiff with exodiff_dg.cfg: ELEMENT VARIABLES relative
// 1e-7, tests/regression/inciter/compflow/Euler/*/exodiff_dg.cfg).
//
//   reference: tk::ExodusIIMeshWriter (src/IO/ExodusIIMeshWriter.cpp) on the ExodusII C library
//   (a third-party library absent here) -- writeMesh / writeElemVarNames / writeTimeStamp /
//   writeElemScalar as DG::writeFields drives them (src/Inciter/DG.cpp:1165-1215,
//   Discretization::write).
//   here: the file format itself (netCDF classic: big-endian header of dimensions, attributes,
//   variables; fixed-size data; one record per time step) written directly, with the dimension,
//   variable and attribute names of the ExodusII conventions: one TETRA element block, side
//   sets as (element, ExodusII side number) pairs, element variables vals_elem_var<V>eb1.
// Host I/O; no device code.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/qdg.h"
#include "qdg_host.hpp"

using namespace qdg;

namespace {

enum { NC_BYTE = 1, NC_CHAR = 2, NC_INT = 4, NC_DOUBLE = 6 };
enum { T_DIM = 0x0A, T_VAR = 0x0B, T_ATT = 0x0C };

struct Out {
  std::vector<unsigned char> b;
  void i32(int32_t v) { for (int s = 24; s >= 0; s -= 8) b.push_back((unsigned char)((uint32_t)v >> s)); }
  void i64(int64_t v) { for (int s = 56; s >= 0; s -= 8) b.push_back((unsigned char)((uint64_t)v >> s)); }
  void f64(double v) { uint64_t u; std::memcpy(&u, &v, 8); i64((int64_t)u); }
  void pad() { while (b.size() % 4) b.push_back(0); }
  void name(const std::string& s) { i32((int32_t)s.size()); b.insert(b.end(), s.begin(), s.end()); pad(); }
};

struct Att { std::string name; int type; std::string text; std::vector<int32_t> ints; };
struct Var {
  std::string name; std::vector<int> dims; int type; std::vector<Att> atts;
  bool rec = false; size_t vsize = 0; int64_t begin = 0;
  std::vector<unsigned char> data;       // fixed variables: the whole (padded) payload
};

size_t tsize(int t) { return t == NC_DOUBLE ? 8 : t == NC_INT ? 4 : 1; }

void put_att(Out& o, const Att& a)
{
  o.name(a.name);
  o.i32(a.type);
  if (a.type == NC_CHAR) { o.i32((int32_t)a.text.size()); o.b.insert(o.b.end(), a.text.begin(), a.text.end()); o.pad(); }
  else { o.i32((int32_t)a.ints.size()); for (int32_t v : a.ints) o.i32(v); }
}

void be_ints(std::vector<unsigned char>& d, const std::vector<int32_t>& v)
{
  Out o; for (int32_t x : v) o.i32(x); d = o.b;
}
void be_doubles(std::vector<unsigned char>& d, const double* v, size_t n)
{
  Out o; o.b.reserve(8 * n); for (size_t i = 0; i < n; ++i) o.f64(v[i]); d = o.b;
}
void chars(std::vector<unsigned char>& d, const std::vector<std::string>& names, size_t len)
{
  d.assign(names.size() * len, 0);
  for (size_t i = 0; i < names.size(); ++i) std::memcpy(d.data() + i * len, names[i].data(), std::min(len - 1, names[i].size()));
  while (d.size() % 4) d.push_back(0);
}

}  // namespace

extern "C" int qdg_exo_write(const char* path, const char* title, size_t nnode, const double* x, const double* y,
                             const double* z, size_t nelem, const size_t* inpoel, size_t nss,
                             const int32_t* ss_id, const size_t* ss_off, const size_t* ss_elem,
                             const int32_t* ss_side, size_t nvar, const char* const* var_names, size_t ntime,
                             const double* times, const double* vals)
{
  QDG_TRY
  if (!path || !x || !y || !z || !inpoel) return fail("qdg_exo_write: null argument");
  if (nss && (!ss_id || !ss_off || !ss_elem || !ss_side)) return fail("qdg_exo_write: null side-set arrays");
  if (ntime && !times) return fail("qdg_exo_write: null times");          // written for every record, fields or not
  if (nvar && (!var_names || (ntime && !vals))) return fail("qdg_exo_write: null field arrays");
  if (nelem > (size_t)INT32_MAX / 4 || nnode > (size_t)INT32_MAX) return fail("qdg_exo_write: mesh too large for 32-bit ids");
  const size_t LEN_STRING = 33, LEN_NAME = 33;
  // ---- dimensions --------------------------------------------------------------------------
  std::vector<std::pair<std::string, int32_t>> dims;
  auto dim = [&](const std::string& n, size_t len) { dims.emplace_back(n, (int32_t)len); return (int)dims.size() - 1; };
  const int d_lenstr = dim("len_string", LEN_STRING), d_lenline = dim("len_line", 81), d_four = dim("four", 4);
  const int d_lenname = dim("len_name", LEN_NAME), d_time = dim("time_step", 0), d_ndim = dim("num_dim", 3);
  const int d_nnode = dim("num_nodes", nnode), d_nelem = dim("num_elem", nelem), d_nblk = dim("num_el_blk", 1);
  const int d_neb = dim("num_el_in_blk1", nelem), d_npe = dim("num_nod_per_el1", 4);
  (void)d_lenstr; (void)d_lenline; (void)d_four; (void)d_nelem;
  int d_nss = -1, d_nev = -1;
  std::vector<int> d_ss(nss, -1);
  if (nss) {
    d_nss = dim("num_side_sets", nss);
    for (size_t k = 0; k < nss; ++k) d_ss[k] = dim("num_side_ss" + std::to_string(k + 1), ss_off[k + 1] - ss_off[k]);
  }
  if (nvar) d_nev = dim("num_elem_var", nvar);
  // ---- variables ----------------------------------------------------------------------------
  std::vector<Var> vars;
  auto add = [&](const std::string& n, std::vector<int> dd, int type) -> Var& {
    vars.push_back(Var{ n, std::move(dd), type, {}, false, 0, 0, {} });
    Var& v = vars.back();
    v.rec = !v.dims.empty() && v.dims[0] == d_time;
    return v;
  };
  { Var& v = add("time_whole", { d_time }, NC_DOUBLE); (void)v; }
  { Var& v = add("eb_status", { d_nblk }, NC_INT); be_ints(v.data, { 1 }); }
  { Var& v = add("eb_prop1", { d_nblk }, NC_INT); v.atts.push_back({ "name", NC_CHAR, "ID", {} }); be_ints(v.data, { 1 }); }
  if (nss) {
    { Var& v = add("ss_status", { d_nss }, NC_INT); be_ints(v.data, std::vector<int32_t>(nss, 1)); }
    { Var& v = add("ss_prop1", { d_nss }, NC_INT); v.atts.push_back({ "name", NC_CHAR, "ID", {} });
      be_ints(v.data, std::vector<int32_t>(ss_id, ss_id + nss)); }
  }
  { Var& v = add("coordx", { d_nnode }, NC_DOUBLE); be_doubles(v.data, x, nnode); }
  { Var& v = add("coordy", { d_nnode }, NC_DOUBLE); be_doubles(v.data, y, nnode); }
  { Var& v = add("coordz", { d_nnode }, NC_DOUBLE); be_doubles(v.data, z, nnode); }
  { Var& v = add("coor_names", { d_ndim, d_lenname }, NC_CHAR); chars(v.data, { "x", "y", "z" }, LEN_NAME); }
  {
    Var& v = add("connect1", { d_neb, d_npe }, NC_INT);
    v.atts.push_back({ "elem_type", NC_CHAR, "TETRA", {} });
    std::vector<int32_t> c(4 * nelem);
    for (size_t i = 0; i < 4 * nelem; ++i) {
      if (inpoel[i] >= nnode) return fail("qdg_exo_write: inpoel entry out of range");
      c[i] = (int32_t)inpoel[i] + 1;                      // ExodusII ids start at 1
    }
    be_ints(v.data, c);
  }
  for (size_t k = 0; k < nss; ++k) {
    const size_t n = ss_off[k + 1] - ss_off[k];
    std::vector<int32_t> e(n), sd(n);
    for (size_t i = 0; i < n; ++i) {
      if (ss_elem[ss_off[k] + i] >= nelem) return fail("qdg_exo_write: side-set element out of range");
      e[i] = (int32_t)ss_elem[ss_off[k] + i] + 1; sd[i] = ss_side[ss_off[k] + i];
    }
    { Var& v = add("elem_ss" + std::to_string(k + 1), { d_ss[k] }, NC_INT); be_ints(v.data, e); }
    { Var& v = add("side_ss" + std::to_string(k + 1), { d_ss[k] }, NC_INT); be_ints(v.data, sd); }
  }
  if (nvar) {
    std::vector<std::string> nm(nvar);
    for (size_t i = 0; i < nvar; ++i) nm[i] = var_names[i] ? var_names[i] : "";
    { Var& v = add("name_elem_var", { d_nev, d_lenname }, NC_CHAR); chars(v.data, nm, LEN_NAME); }
    { Var& v = add("elem_var_tab", { d_nblk, d_nev }, NC_INT); be_ints(v.data, std::vector<int32_t>(nvar, 1)); }
    for (size_t i = 0; i < nvar; ++i) add("vals_elem_var" + std::to_string(i + 1) + "eb1", { d_time, d_neb }, NC_DOUBLE);
  }
  // ---- sizes and offsets ----------------------------------------------------------------------
  auto prod = [&](const Var& v, size_t from) { size_t n = 1; for (size_t i = from; i < v.dims.size(); ++i) n *= (size_t)dims[v.dims[i]].second; return n; };
  for (Var& v : vars) {
    const size_t n = v.rec ? prod(v, 1) : prod(v, 0);
    v.vsize = (n * tsize(v.type) + 3) / 4 * 4;
    if (!v.rec) v.data.resize(v.vsize, 0);
  }
  std::vector<Att> gatts = { { "api_version", NC_DOUBLE, "", {} }, { "version", NC_DOUBLE, "", {} },
                            { "floating_point_word_size", NC_INT, "", { 8 } }, { "file_size", NC_INT, "", { 1 } },
                            { "int64_status", NC_INT, "", { 0 } },
                            { "title", NC_CHAR, title ? title : "qdg", {} } };
  auto header = [&](Out& o) {
    o.b = { 'C', 'D', 'F', 2 };
    o.i32((int32_t)ntime);
    o.i32(T_DIM); o.i32((int32_t)dims.size());
    for (auto& d : dims) { o.name(d.first); o.i32(d.second); }
    o.i32(T_ATT); o.i32((int32_t)gatts.size());
    for (const Att& a : gatts) {
      if (a.type == NC_DOUBLE) { o.name(a.name); o.i32(NC_DOUBLE); o.i32(1); o.f64(a.name == "api_version" ? 7.0 : 7.0); }
      else put_att(o, a);
    }
    o.i32(T_VAR); o.i32((int32_t)vars.size());
    for (const Var& v : vars) {
      o.name(v.name);
      o.i32((int32_t)v.dims.size());
      for (int d : v.dims) o.i32(d);
      if (v.atts.empty()) { o.i32(0); o.i32(0); }
      else { o.i32(T_ATT); o.i32((int32_t)v.atts.size()); for (const Att& a : v.atts) put_att(o, a); }
      o.i32(v.type);
      o.i32((int32_t)std::min<size_t>(v.vsize, 0x7fffffffu));
      o.i64(v.begin);
    }
  };
  Out probe;
  header(probe);                                   // header size does not depend on the offsets
  int64_t off = (int64_t)probe.b.size();
  for (Var& v : vars) if (!v.rec) { v.begin = off; off += (int64_t)v.vsize; }
  size_t recsize = 0;
  for (Var& v : vars) if (v.rec) { v.begin = off + (int64_t)recsize; recsize += v.vsize; }
  Out h;
  header(h);
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail(std::string("qdg_exo_write: cannot open ") + path);
  bool ok = std::fwrite(h.b.data(), 1, h.b.size(), f) == h.b.size();
  for (const Var& v : vars) if (!v.rec) ok = ok && std::fwrite(v.data.data(), 1, v.data.size(), f) == v.data.size();
  std::vector<unsigned char> rec;
  for (size_t t = 0; t < ntime && ok; ++t) {
    Out o;
    o.b.reserve(recsize);
    size_t ivar = 0;
    for (const Var& v : vars) {
      if (!v.rec) continue;
      if (v.name == "time_whole") { o.f64(times[t]); continue; }
      const double* src = vals + (t * nvar + ivar) * nelem;     // vals[time][var][elem]
      for (size_t e = 0; e < nelem; ++e) o.f64(src[e]);
      ++ivar;
    }
    ok = std::fwrite(o.b.data(), 1, o.b.size(), f) == o.b.size();
  }
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) return fail(std::string("qdg_exo_write: write error on ") + path);
  return 0;
  QDG_CATCH
}
