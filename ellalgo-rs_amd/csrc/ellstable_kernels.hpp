// ellstable_kernels.hpp -- kernels for EllStable::update_core (src/ell_stable.rs:52-125).
#pragma once

#include <hip/hip_runtime.h>

#include "ell_kernels.hpp"

namespace ellhip {
}  // namespace ellhip
