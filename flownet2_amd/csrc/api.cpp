// Version / error plumbing of libflownet2_hip.so.
#include "fn2_common.hpp"

namespace fn2 {
std::string& last_error() {
  static thread_local std::string e;
  return e;
}
int& batch_invariant_flag() {
  static int on = 0;
  return on;
}
}  // namespace fn2

FN2_API const char* fn2_version(void) { return "0.1 (gfx950)"; }
FN2_API const char* fn2_last_error_string(void) { return fn2::last_error().c_str(); }
FN2_API int fn2_set_batch_invariant(int on) { fn2::batch_invariant_flag() = on ? 1 : 0; return FN2_OK; }
FN2_API int fn2_get_batch_invariant(void) { return fn2::batch_invariant_flag(); }
