// stands in for <hip/hip_runtime.h> when the kernel sources are compiled for the CPU simulator (tests/hipsim/hipsim.h)
#pragma once
#include "../hipsim.h"
