// third-party stand-in (compile-only check) for glog: stream-style LOG / CHECK macros with glog's names and shapes
#pragma once
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
namespace fn2_glog {
class Msg {
 public:
  explicit Msg(bool fatal) : fatal_(fatal) {}
  ~Msg() { if (fatal_) { std::cerr << s_.str() << std::endl; std::abort(); } }
  std::ostream& stream() { return s_; }
 private:
  std::ostringstream s_;
  bool fatal_;
};
struct Voidify { void operator&(std::ostream&) {} };
}  // namespace fn2_glog
#define FN2_GLOG_INFO ::fn2_glog::Msg(false).stream()
#define FN2_GLOG_WARNING ::fn2_glog::Msg(false).stream()
#define FN2_GLOG_ERROR ::fn2_glog::Msg(false).stream()
#define FN2_GLOG_FATAL ::fn2_glog::Msg(true).stream()
#define LOG(sev) FN2_GLOG_##sev
#define DLOG(sev) LOG(sev)
#define LOG_IF(sev, cond) !(cond) ? (void)0 : ::fn2_glog::Voidify() & LOG(sev)
#define LOG_FIRST_N(sev, n) LOG(sev)
#define CHECK(cond) (cond) ? (void)0 : ::fn2_glog::Voidify() & ::fn2_glog::Msg(true).stream() << "Check failed: " #cond " "
#define FN2_GLOG_OP(a, b, op) ((a) op (b)) ? (void)0 : ::fn2_glog::Voidify() & ::fn2_glog::Msg(true).stream() << "Check failed: " #a " " #op " " #b " "
#define CHECK_EQ(a, b) FN2_GLOG_OP(a, b, ==)
#define CHECK_NE(a, b) FN2_GLOG_OP(a, b, !=)
#define CHECK_LE(a, b) FN2_GLOG_OP(a, b, <=)
#define CHECK_LT(a, b) FN2_GLOG_OP(a, b, <)
#define CHECK_GE(a, b) FN2_GLOG_OP(a, b, >=)
#define CHECK_GT(a, b) FN2_GLOG_OP(a, b, >)
#define CHECK_NOTNULL(p) (p)
#define DCHECK(c) CHECK(c)
#define DCHECK_EQ(a, b) CHECK_EQ(a, b)
#define DCHECK_NE(a, b) CHECK_NE(a, b)
#define DCHECK_GT(a, b) CHECK_GT(a, b)
#define DCHECK_LT(a, b) CHECK_LT(a, b)
#define DCHECK_LE(a, b) CHECK_LE(a, b)
#define DCHECK_GE(a, b) CHECK_GE(a, b)
