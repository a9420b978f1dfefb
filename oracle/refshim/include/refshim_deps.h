// ORACLE — TEST INFRASTRUCTURE ONLY.  What the drivers (ref_*.cpp) include for "the third-party world": here the stand-ins; the twin of
// this file under include_real/ pulls in the REAL Eigen / PCL / Ceres instead (oracle/refshim/Makefile REAL_DEPS=1, README.md).
#pragma once
#include "refshim/ros_pcl_min.h"
#include "refshim/ceres_min.h"
