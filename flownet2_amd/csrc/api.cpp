// Version / error plumbing of libflownet2_hip.so.
#include "fn2_common.hpp"

namespace fn2 {
std::string& last_error() {
  static thread_local std::string e;
  return e;
}
}  // namespace fn2

FN2_API const char* fn2_version(void) { return "0.1 (gfx950)"; }
FN2_API const char* fn2_last_error_string(void) { return fn2::last_error().c_str(); }
