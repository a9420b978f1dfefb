// stand-in: ROS assertion macros are not used on the paths ref_imu.cpp drives
#pragma once
