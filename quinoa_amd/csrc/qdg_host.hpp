// qdg_host.hpp -- host-side internals of libqdg (error channel, handles).
#pragma once
#include <cstddef>
#include <cstdint>
#include <exception>
#include <memory>
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

struct qdg_refined {
  size_t nnode = 0;
  rawvec<size_t> inpoel, parent, tri;
  rawvec<double> x, y, z;
};

