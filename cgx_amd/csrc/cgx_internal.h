/* cgx_internal.h -- symbols shared between the host C and the device translation units, not part of the public ABI */
#ifndef CGX_INTERNAL_H
#define CGX_INTERNAL_H
#include "../../include/cgx.h"
#ifdef __cplusplus
extern "C" {
#endif
void cgx__set_host_ms(cgx_ctx *ctx, const char *name, double ms);
#ifdef __cplusplus
}
#endif
#endif
