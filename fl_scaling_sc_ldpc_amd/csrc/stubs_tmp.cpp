#include "common.h"
extern "C" int scldpc_sample_philox_device(const scldpc_code_params *, uint64_t, uint64_t, int32_t, double, int32_t,
                                           const int32_t *, int32_t *, uint32_t *, void *)
{ return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "not implemented yet"); }
extern "C" int scldpc_sw_bp_device(const scldpc_code_params *, int32_t, const int32_t *, const uint32_t *, int32_t,
                                   int32_t, int32_t, int32_t *, uint32_t *, void *)
{ return scldpc::set_error(SCLDPC_ERR_BAD_ARG, "not implemented yet"); }
