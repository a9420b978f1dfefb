#pragma once
#include "../../include/refshim/ceres_min.h"
