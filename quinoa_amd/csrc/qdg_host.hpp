// qdg_host.hpp -- host-side internals of libqdg (error channel, handles).
#pragma once
#include <cstddef>
#include <cstdint>
#include <exception>
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
