// stand-in header (oracle/refshim/README.md)
#pragma once
#include "refshim/ceres_min.h"
