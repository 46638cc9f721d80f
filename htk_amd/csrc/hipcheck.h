// hipcheck.h -- turn a failing HIP runtime call into an HTKAMD_EHIP return with a message.
#ifndef HTKAMD_HIPCHECK_H
#define HTKAMD_HIPCHECK_H
#include <hip/hip_runtime.h>
#include "internal.h"

#define HIPCHECK(call)                                                                         \
   do {                                                                                        \
      hipError_t e_ = (call);                                                                  \
      if (e_ != hipSuccess) {                                                                  \
         htkamd_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
         return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? HTKAMD_ENODEV : HTKAMD_EHIP; \
      }                                                                                        \
   } while (0)

#endif
