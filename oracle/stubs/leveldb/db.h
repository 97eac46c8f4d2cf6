// empty stand-in: LevelDB is named by custom_data_layer.cpp but never used (LOG(FATAL) on that backend)
