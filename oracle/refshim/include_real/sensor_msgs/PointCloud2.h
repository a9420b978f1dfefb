// REAL_DEPS build (oracle/refshim/README.md): ROS plumbing stays a stand-in, everything else is the real library
#pragma once
#include "refshim_deps.h"
