// inst_4096.hip — explicit instantiations of one group of geometries (see modp_instances.h)
#include "modp_instances.h"
VMN_UNIT_4096(template)
