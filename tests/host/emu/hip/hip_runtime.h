// stands in for <hip/hip_runtime.h> when the device headers are compiled for the host (tests/host/emu/wave_emu.hpp)
#pragma once
#include "../wave_emu.hpp"
