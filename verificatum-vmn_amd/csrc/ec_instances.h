// ec_instances.h — the curve kernels of one field size as explicit instantiations (see modp_instances.h: the host unit
// vmnhip.hip declares them `extern template`, csrc/inst_p224.hip / inst_p256.hip / inst_p384.hip / inst_p521.hip define them, compiled side by side).
#pragma once
#include "ec_kernels.h"

#define VMN_EC_INSTANCES(KW, S_, NW_)                                                                                                   \
    KW __global__ void vmn::k_ec_import<S_, NW_>(vmn::u32*, const uint8_t*, size_t, size_t, int, size_t, vmn::ECDev, vmn::u32*);         \
    KW __global__ void vmn::k_ec_export<S_, NW_>(uint8_t*, size_t, size_t, int, const vmn::u32*, size_t, vmn::ECDev);                    \
    KW __global__ void vmn::k_ec_add<S_>(vmn::u32*, const vmn::u32*, const vmn::u32*, size_t, size_t, vmn::ECDev);                       \
    KW __global__ void vmn::k_ec_neg<S_>(vmn::u32*, const vmn::u32*, size_t, vmn::ECDev);                                                \
    KW __global__ void vmn::k_ec_equal<S_>(const vmn::u32*, const vmn::u32*, size_t, vmn::ECDev, vmn::u32*);                             \
    KW __global__ void vmn::k_ec_mulvar<S_>(vmn::u32*, const vmn::u32*, const vmn::u32*, int, size_t, int, int, size_t, vmn::ECDev,      \
                                            vmn::u32*);                                                                                 \
    KW __global__ void vmn::k_ec_mulvar2<S_>(vmn::u32*, const vmn::u32*, const vmn::u32*, int, int, const vmn::u32*, const vmn::u32*, int,  \
                                             size_t, int, int, size_t, vmn::ECDev, vmn::u32*);                                          \
    KW __global__ void vmn::k_ec_chain<S_>(vmn::u32*, const vmn::u32*, int, vmn::ECDev);                                                 \
    KW __global__ void vmn::k_ec_fixed_level<S_>(vmn::u32*, int, int, int, vmn::ECDev);                                                  \
    KW __global__ void vmn::k_ec_fixed_exp<S_>(vmn::u32*, const vmn::u32*, int, int, const vmn::u32*, int, size_t, vmn::ECDev);          \
    KW __global__ void vmn::k_finv_up<S_, true>(vmn::u32*, vmn::u32*, vmn::LevelInputs, unsigned, size_t, size_t, vmn::ECDev);           \
    KW __global__ void vmn::k_finv_up<S_, false>(vmn::u32*, vmn::u32*, vmn::LevelInputs, unsigned, size_t, size_t, vmn::ECDev);          \
    KW __global__ void vmn::k_finv_top<S_>(vmn::u32*, const vmn::u32*, size_t, vmn::ECDev);                                              \
    KW __global__ void vmn::k_finv_down<S_>(vmn::u32*, const vmn::u32*, const vmn::u32*, const vmn::u32*, size_t, size_t, vmn::ECDev);   \
    KW __global__ void vmn::k_ec_normalize_down<S_>(vmn::u32*, vmn::LevelInputs, unsigned, const vmn::u32*, const vmn::u32*, size_t,     \
                                                    size_t, vmn::ECDev);                                                                \
    KW __global__ void vmn::k_ec_bucket_level<S_, true>(vmn::u32*, size_t, vmn::LevelInputs, unsigned, const vmn::u32*, const vmn::u32*, \
                                                        const vmn::u32*, const vmn::u32*, size_t, size_t, vmn::u32, vmn::ECDev);        \
    KW __global__ void vmn::k_ec_bucket_level<S_, false>(vmn::u32*, size_t, vmn::LevelInputs, unsigned, const vmn::u32*, const vmn::u32*, \
                                                         const vmn::u32*, const vmn::u32*, size_t, size_t, vmn::u32, vmn::ECDev);       \
    KW __global__ void vmn::k_ec_bucket_first_jacobian<S_>(vmn::u32*, size_t, vmn::LevelInputs, unsigned, const vmn::u32*, const vmn::u32*, \
                                                           const vmn::u32*, const vmn::u32*, size_t, size_t, vmn::u32, vmn::ECDev);     \
    KW __global__ void vmn::k_ec_reduce<S_>(vmn::u32*, const vmn::u32*, size_t, size_t, size_t, vmn::ECDev);                             \
    KW __global__ void vmn::k_ec_scan_totals<S_>(vmn::u32*, const vmn::u32*, size_t, size_t, size_t, int, vmn::ECDev);                   \
    KW __global__ void vmn::k_ec_scan_apply<S_>(vmn::u32*, const vmn::u32*, const vmn::u32*, size_t, size_t, size_t, int, vmn::ECDev);   \
    KW __global__ void vmn::k_ec_horner<S_>(vmn::u32*, const vmn::u32*, int, int, int, vmn::ECDev);

#define VMN_UNIT_P224(KW) VMN_EC_INSTANCES(KW, 9, 7)
#define VMN_UNIT_P256(KW) VMN_EC_INSTANCES(KW, 10, 8)
#define VMN_UNIT_P384(KW) VMN_EC_INSTANCES(KW, 15, 12)
#define VMN_UNIT_P521(KW) VMN_EC_INSTANCES(KW, 21, 17)
