// qdg_host.hpp -- host-side internals of libqdg (error channel, handles).
#pragma once
#include <cstddef>
#include <cstdint>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <string>
#include <vector>

namespace qdg {

extern const int LPOFA[4][3];

// error channel of the C ABI: nothing ever throws across it
// (reference behaviour: tk::Exception, src/Base/Exception.hpp:33-57; the
// adapter turns a non-zero status back into Throw(qdg_last_error()))
int fail(const std::string& msg);
void set_error(const std::string& msg);

#define QDG_TRY try {
#define QDG_CATCH                                                        \
  } catch (const std::exception& ex) {                                   \
    return ::qdg::fail(std::string("exception: ") + ex.what());          \
  } catch (...) {                                                        \
    return ::qdg::fail("unknown exception");                             \
  }

}  // namespace qdg

// large work and result arrays: no value-initialisation on resize, so that their pages are first
// touched by the threads that fill them (a serial zero-fill of a few hundred MB costs more than
// the refinement itself)
template <class T> struct raw_alloc : std::allocator<T> {
  template <class U> struct rebind { using other = raw_alloc<U>; };
  raw_alloc() = default;
  template <class U> raw_alloc(const raw_alloc<U>&) {}
  template <class U, class... A> void construct(U* p, A&&... a)
  {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
};
template <class T> using rawvec = std::vector<T, raw_alloc<T>>;

// a host thread that is still filling a qdg_refined from device buffers (qdg_mesh_refine_uniform): joined by
// whoever needs the data or wants to free the buffers it reads
struct qdg_host_copy {
  std::thread th;
  std::mutex mu;
  std::string error;
  void join() { std::lock_guard<std::mutex> g(mu); if (th.joinable()) th.join(); }
  ~qdg_host_copy() { join(); }
};

struct qdg_refined {
  size_t nnode = 0;
  rawvec<size_t> inpoel, parent, tri;
  rawvec<double> x, y, z;
  std::vector<int32_t> tri_set;                 // side set of every refined triangle (qdg_mesh_refine_uniform)
  std::shared_ptr<qdg_host_copy> pending;       // the arrays above are complete once this has been joined
  void wait() const { if (pending) pending->join(); }
};


// one rank's uniformly refined chunk (qdg_refine_chunk on the host, qdg_mesh_refine_chunk from the device)
struct qdg_chunk_refined {
  size_t nielem = 0, nunk = 0, nnode = 0;
  rawvec<size_t> inpoel, gid, parent;
  std::vector<size_t> tri, send_off, send_list, recv_counts;
  std::vector<int32_t> tri_set;
  rawvec<double> x, y, z;
  // the plan's entries when they can differ from the caller's (two ghost layers, qdg_refine_chunk_depth)
  std::vector<int32_t> nbr_rank, nbr_layer;
  size_t nghost1 = 0;
};
