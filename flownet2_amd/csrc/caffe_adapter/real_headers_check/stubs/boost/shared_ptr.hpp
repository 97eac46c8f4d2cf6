// third-party stand-in for the compile-only check against the reference's real Caffe headers: boost::shared_ptr = std::shared_ptr
#pragma once
#include <memory>
namespace boost { using std::shared_ptr; class mutex; class thread; }
