// modp_instances.h — the kernels of one execution geometry as explicit instantiations.
//
// Every kernel of modp_kernels.h is a template over Cfg<S, LPE>; ten geometries x sixteen kernels in ONE translation unit
// took hipcc four minutes.  The host code (vmnhip.hip) therefore only DECLARES them (`extern template`) and the
// instantiation units csrc/inst_*.hip define them, a few geometries each, compiled side by side by
// __graft_entry__.build().  A kernel's host stub is an ordinary external symbol, its device code lives in the code
// object of the unit that instantiated it; nothing calls across units on the device.
#pragma once
#include "modp_kernels.h"
#include "modp_shared_exp.h"

// KW = `extern template` (declaration) or `template` (definition)
#define VMN_MODP_INSTANCES(KW, S_, NW_, LPE_)                                                                                          \
    KW __global__ void vmn::k_import_be<vmn::Cfg<S_, LPE_>, NW_>(vmn::u32*, const uint8_t*, size_t, size_t, int, size_t, const vmn::u32*, \
                                                                 vmn::u32, const vmn::u32*, vmn::u32*);                                   \
    KW __global__ void vmn::k_export_be<vmn::Cfg<S_, LPE_>, NW_>(uint8_t*, size_t, size_t, int, const vmn::u32*, size_t, const vmn::u32*, \
                                                                 vmn::u32);                                                               \
    KW __global__ void vmn::k_to_words<vmn::Cfg<S_, LPE_>, NW_>(vmn::u32*, const vmn::u32*, size_t, const vmn::u32*, vmn::u32);           \
    KW __global__ void vmn::k_mul<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, const vmn::u32*, size_t, size_t, const vmn::u32*,      \
                                                      vmn::u32);                                                                         \
    KW __global__ void vmn::k_modpow<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, const vmn::u32*, int, size_t, int, int, size_t,     \
                                                         const vmn::u32*, vmn::u32, const vmn::u32*, vmn::u32*);                         \
    KW __global__ void vmn::k_modpow_phased<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, const vmn::u32*, int, size_t, int, int,      \
                                                                size_t, const vmn::u32*, vmn::u32, const vmn::u32*, vmn::u32*, int,      \
                                                                vmn::u32*, vmn::u32*);                                                   \
    KW __global__ void vmn::k_modpow2<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, const vmn::u32*, int, size_t, int, const vmn::u32*,  \
                                                          const vmn::u32*, int, size_t, int, int, size_t, const vmn::u32*, vmn::u32,      \
                                                          const vmn::u32*, vmn::u32*);                                                  \
    KW __global__ void vmn::k_modpow2_phased<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, const vmn::u32*, int, size_t, int,          \
                                                                 const vmn::u32*, const vmn::u32*, int, size_t, int, int, size_t,        \
                                                                 const vmn::u32*, vmn::u32, const vmn::u32*, vmn::u32*, int, vmn::u32*,  \
                                                                 vmn::u32*);                                                            \
    KW __global__ void vmn::k_modpow_shared<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, const vmn::SlideStep*, int, int, size_t,      \
                                                                const vmn::u32*, vmn::u32, vmn::u32*);                                  \
    KW __global__ void vmn::k_modpow_shared_phased<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, const vmn::SlideStep*, int, int,      \
                                                                       size_t, const vmn::u32*, vmn::u32, vmn::u32*, int, vmn::u32*,     \
                                                                       vmn::u32*);                                                      \
    KW __global__ void vmn::k_modpow_jobs<vmn::Cfg<S_, LPE_>>(vmn::ModpowJob, vmn::ModpowJob, unsigned, int, const vmn::u32*, vmn::u32,    \
                                                              const vmn::u32*, vmn::u32*);                                             \
    KW __global__ void vmn::k_reduce_strided<vmn::Cfg<S_, LPE_>, true>(vmn::u32*, const vmn::u32*, size_t, size_t, size_t,               \
                                                                       const vmn::u32*, vmn::u32);                                       \
    KW __global__ void vmn::k_reduce_strided<vmn::Cfg<S_, LPE_>, false>(vmn::u32*, const vmn::u32*, size_t, size_t, size_t,              \
                                                                        const vmn::u32*, vmn::u32);                                      \
    KW __global__ void vmn::k_ring_elementwise<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, const vmn::u32*, const vmn::u32*, int,    \
                                                                   size_t, const vmn::u32*, vmn::u32);                                   \
    KW __global__ void vmn::k_scan_totals<vmn::Cfg<S_, LPE_>, true>(vmn::u32*, const vmn::u32*, const vmn::u32*, size_t, size_t, size_t, \
                                                                    int, const vmn::u32*, vmn::u32, const vmn::u32*);                    \
    KW __global__ void vmn::k_scan_totals<vmn::Cfg<S_, LPE_>, false>(vmn::u32*, const vmn::u32*, const vmn::u32*, size_t, size_t, size_t, \
                                                                     int, const vmn::u32*, vmn::u32, const vmn::u32*);                   \
    KW __global__ void vmn::k_scan_apply<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, const vmn::u32*, const vmn::u32*, size_t,       \
                                                             size_t, size_t, int, const vmn::u32*, vmn::u32, const vmn::u32*);           \
    KW __global__ void vmn::k_fixed_level<vmn::Cfg<S_, LPE_>>(vmn::u32*, int, int, int, const vmn::u32*, vmn::u32);                      \
    KW __global__ void vmn::k_fixed_exp<vmn::Cfg<S_, LPE_>>(vmn::u32*, const vmn::u32*, int, int, const vmn::u32*, int, size_t, int,     \
                                                            const vmn::u32*, vmn::u32);                                                  \
    KW __global__ void vmn::k_bucket_level<vmn::Cfg<S_, LPE_>, true>(vmn::u32*, size_t, vmn::LevelInputs, unsigned, const vmn::u32*,     \
                                                                     const vmn::u32*, const vmn::u32*, const vmn::u32*, size_t, size_t, \
                                                                     vmn::u32, const vmn::u32*, vmn::u32);                               \
    KW __global__ void vmn::k_bucket_level<vmn::Cfg<S_, LPE_>, false>(vmn::u32*, size_t, vmn::LevelInputs, unsigned, const vmn::u32*,    \
                                                                      const vmn::u32*, const vmn::u32*, const vmn::u32*, size_t, size_t, \
                                                                      vmn::u32, const vmn::u32*, vmn::u32);

// subgroup membership: one element per lane (LPE = 1) or the element's own lanes (LPE > 1, base geometries only)
#define VMN_MEMBER_INSTANCE_ONE_LANE(KW, S_, LPE_) \
    KW __global__ void vmn::k_jacobi_member<vmn::Cfg<S_, LPE_>>(const vmn::u32*, size_t, const vmn::u32*, vmn::u32*);
#define VMN_MEMBER_INSTANCE_LANES(KW, S_, LPE_) \
    KW __global__ void vmn::k_jacobi_member_lanes<vmn::Cfg<S_, LPE_>>(const vmn::u32*, size_t, const vmn::u32*, vmn::u32*);

// The geometries (limbs, packed words, lanes per element; see VMN_FOR_SIZES in vmnhip.hip), grouped into the instantiation units
#define VMN_UNIT_SMALL(KW)                                                                                             \
    VMN_MODP_INSTANCES(KW, 10, 8, 1) VMN_MODP_INSTANCES(KW, 14, 12, 1) VMN_MODP_INSTANCES(KW, 19, 16, 1)                 \
    VMN_MODP_INSTANCES(KW, 37, 32, 1) VMN_MEMBER_INSTANCE_ONE_LANE(KW, 10, 1) VMN_MEMBER_INSTANCE_ONE_LANE(KW, 14, 1)     \
    VMN_MEMBER_INSTANCE_ONE_LANE(KW, 19, 1) VMN_MEMBER_INSTANCE_ONE_LANE(KW, 37, 1)
#define VMN_UNIT_2048(KW) VMN_MODP_INSTANCES(KW, 74, 64, 1) VMN_MEMBER_INSTANCE_ONE_LANE(KW, 74, 1)
#define VMN_UNIT_2048_WIDE(KW)                                                                                         \
    VMN_MODP_INSTANCES(KW, 76, 64, 4) VMN_MODP_INSTANCES(KW, 80, 64, 8)                                                \
    KW __global__ void vmn::k_modpow_jobs_mixed<vmn::Cfg<80, 8>, vmn::Cfg<76, 4>>(vmn::ModpowJob, vmn::ModpowJob, unsigned, int,       \
                                                                                 const vmn::u32*, vmn::u32, const vmn::u32*, vmn::u32*);
#define VMN_UNIT_3072(KW) VMN_MODP_INSTANCES(KW, 110, 96, 2) VMN_MODP_INSTANCES(KW, 112, 96, 4) VMN_MEMBER_INSTANCE_LANES(KW, 110, 2)
#define VMN_UNIT_4096(KW) VMN_MODP_INSTANCES(KW, 148, 128, 4) VMN_MEMBER_INSTANCE_LANES(KW, 148, 4)
#define VMN_UNIT_8192(KW) VMN_MODP_INSTANCES(KW, 296, 256, 8) VMN_MEMBER_INSTANCE_LANES(KW, 296, 8)
#define VMN_UNIT_16384(KW) VMN_MODP_INSTANCES(KW, 592, 512, 16) VMN_MEMBER_INSTANCE_LANES(KW, 592, 16)
