// Stand-in for the HIP runtime in the host-only sanitizer build of OUR library's argument validation
// (`make -C nwhead_amd/csrc sanitize`): no device exists there, so every runtime entry point the host code can reach
// answers "no device / launch failed".  The validation paths under test return before any of these matters.
#include <cstddef>
extern "C" {
typedef int hipError_t_;
void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
hipError_t_ __hipPushCallConfiguration(...) { return 0; }
hipError_t_ __hipPopCallConfiguration(...) { return 0; }
hipError_t_ hipLaunchKernel(...) { return 98; }                 /* hipErrorInvalidDeviceFunction */
hipError_t_ hipGetLastError(void) { return 98; }
hipError_t_ hipMemsetAsync(...) { return 101; }                 /* hipErrorInvalidDevice */
hipError_t_ hipGetDeviceCount(int* n) { if (n) *n = 0; return 100; }   /* hipErrorNoDevice */
hipError_t_ hipGetDevice(int* d) { if (d) *d = 0; return 100; }
hipError_t_ hipGetDevicePropertiesR0600(...) { return 100; }
hipError_t_ hipDeviceGetAttribute(int* v, ...) { if (v) *v = 0; return 100; }
hipError_t_ hipGetSymbolAddress(void** p, const void*) { if (p) *p = nullptr; return 100; }
hipError_t_ hipFuncSetAttribute(...) { return 100; }
void __hipRegisterVar(void**, void*, char*, const char*, int, unsigned long, int, int) {}
hipError_t_ hipEventCreate(...) { return 100; }
hipError_t_ hipEventRecord(...) { return 100; }
hipError_t_ hipEventSynchronize(...) { return 100; }
hipError_t_ hipEventElapsedTime(...) { return 100; }
}
