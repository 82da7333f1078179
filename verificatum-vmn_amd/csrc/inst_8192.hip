// inst_8192.hip — explicit instantiations of one group of geometries (see modp_instances.h)
#include "modp_instances.h"
VMN_UNIT_8192(template)
