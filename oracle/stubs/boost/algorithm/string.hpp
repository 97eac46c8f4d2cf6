// stand-in: included by custom_data_layer.cpp; nothing from Boost.StringAlgo is used there, but the real header is what brings
// <cassert> into that file
#pragma once
#include <cassert>
