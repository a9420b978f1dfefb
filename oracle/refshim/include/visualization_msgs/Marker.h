// stand-in header (oracle/refshim/README.md): everything lives in refshim/ros_pcl_min.h
#pragma once
#include "refshim/ros_pcl_min.h"
