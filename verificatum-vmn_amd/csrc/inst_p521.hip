// inst_p521.hip — explicit instantiations of the curve kernels over a 21-limb field (20 limbs would leave no padding word for the infinity flag of a row) (P-521; see ec_instances.h)
#include "ec_instances.h"
VMN_UNIT_P521(template)
