// libcdsmvs_hip.so — version entry point.
#include "cds_common.hpp"
extern "C" int cds_version(void) { return 100; /* 0.1.0 */ }
