// ORACLE — TEST INFRASTRUCTURE ONLY.  Compiles the reference's LiLi-OM/src/Preprocessing.cpp UNMODIFIED
// (include path -> /root/reference/LiLi-OM, third-party headers -> oracle/refshim/include) into
// oracle/_ref/libref_livox.so.  No reference source is copied into this repository.
#include "refshim_deps.h"
#define main ref_livox_node_main
#include "src/Preprocessing.cpp"
#undef main
#include "ref_pre_driver.inc"
