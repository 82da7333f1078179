// inst_p384.hip — explicit instantiations of the curve kernels over a 15-limb field (see ec_instances.h)
#include "ec_instances.h"
VMN_UNIT_P384(template)
