// inst_p224.hip — explicit instantiations of the curve kernels over a 9-limb field (P-224; see ec_instances.h)
#include "ec_instances.h"
VMN_UNIT_P224(template)
