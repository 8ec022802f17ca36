// tests/emul: see hip_runtime.h
#include "hip_runtime.h"
