// Stand-in for include/caffe/net.hpp (included by custom_data_layer.cpp, unused there).
#pragma once
#include "caffe/layer.hpp"
