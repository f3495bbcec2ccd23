/* cgx_internal.h -- symbols shared between the host C and the device translation units, not part of the public ABI */
#ifndef CGX_INTERNAL_H
#define CGX_INTERNAL_H
#include "../../include/cgx.h"
#ifdef __cplusplus
extern "C" {
#endif
void cgx__set_host_ms(cgx_ctx *ctx, const char *name, double ms);
int64_t cgx__option(cgx_ctx *ctx, const char *name);          /* options the host side reads ("async_write") */
void cgx__set_host_state(cgx_ctx *ctx, void *p);              /* per-context host state (background writer) */
void *cgx__get_host_state(cgx_ctx *ctx);
void cgx__host_release(cgx_ctx *ctx);
int cgx__host_busy(cgx_ctx *ctx);                             /* batches handed to the writer and not yet joined (implemented by the host TU) */
void cgx__bind_thread(cgx_ctx *ctx);                          /* hipSetDevice(ctx->device) for the calling thread (writer threads) */
const void *cgx__get_vocab_owner(cgx_ctx *ctx);               /* corpus whose spellings / score tables are on the device */
void cgx__set_vocab_owner(cgx_ctx *ctx, const void *corpus);
void cgx__device_cpulist(cgx_ctx *ctx, char *buf, size_t cap);  /* sysfs local_cpulist of the GPU ("0-63,128-191"), "" if unknown */                         /* implemented by the host TU, called from cgx_destroy */
#ifdef __cplusplus
}
#endif
#endif
