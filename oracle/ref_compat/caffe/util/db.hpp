// Stand-in for include/caffe/util/db.hpp (included by custom_data_layer.hpp; the layer talks to LMDB directly).
#pragma once
