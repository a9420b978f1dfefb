// ORACLE — TEST INFRASTRUCTURE ONLY.  Compiles the reference's LiLi-OM-ROT/src/Preprocessing.cpp UNMODIFIED
// (include path -> /root/reference/LiLi-OM-ROT, third-party headers -> oracle/refshim/include) into
// oracle/_ref/libref_rot.so.  No reference source is copied into this repository.
#include "refshim_deps.h"
#define main ref_rot_node_main
#include "src/Preprocessing.cpp"
#undef main
#include "ref_pre_driver.inc"
