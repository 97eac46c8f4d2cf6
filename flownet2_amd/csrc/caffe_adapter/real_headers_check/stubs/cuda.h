// third-party stand-in (compile-only check)
#pragma once
#include "cuda_runtime.h"
