// inst_16384.hip — explicit instantiations of one group of geometries (see modp_instances.h)
#include "modp_instances.h"
VMN_UNIT_16384(template)
