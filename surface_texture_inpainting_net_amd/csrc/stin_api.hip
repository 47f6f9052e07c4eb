// Version / error-string entry points of libstin_hip.so.
#include "stin_common.h"

extern "C" int stin_version(void) { return STIN_VERSION; }

extern "C" const char* stin_error_string(int code) {
    switch (code) {
        case STIN_OK: return "ok";
        case STIN_E_NULL: return "required pointer is NULL";
        case STIN_E_SIZE: return "negative or inconsistent size / leading dimension";
        case STIN_E_ALIGN: return "pointer or leading dimension not aligned as required";
        case STIN_E_WORKSPACE: return "workspace too small";
        case STIN_E_UNSUPPORTED: return "shape or mode not supported by this build";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown stin error";
}
