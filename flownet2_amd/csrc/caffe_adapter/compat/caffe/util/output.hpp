// Stand-in (nothing from this header is used by the hot-path layers).
#pragma once
#include "caffe/common.hpp"
