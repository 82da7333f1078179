// inst_3072.hip — explicit instantiations of one group of geometries (see modp_instances.h)
#include "modp_instances.h"
VMN_UNIT_3072(template)
