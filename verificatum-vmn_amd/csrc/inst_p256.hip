// inst_p256.hip — explicit instantiations of the curve kernels over a 10-limb field (see ec_instances.h)
#include "ec_instances.h"
VMN_UNIT_P256(template)
