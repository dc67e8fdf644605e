// abi.hip — version / error entry points of libllmc_hip.so.
#include "common.h"

namespace llmc {
static thread_local char g_last_error[512] = "";

void set_last_error(const char* where, hipError_t e) {
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}
void set_last_error_msg(const char* msg) { snprintf(g_last_error, sizeof(g_last_error), "%s", msg); }
}  // namespace llmc

extern "C" int llmc_hip_abi_version(void) { return LLMC_HIP_ABI_VERSION; }

extern "C" int llmc_hip_last_error(char* buf_host, size_t n) {
    if (!buf_host || n == 0) return (int)strlen(llmc::g_last_error);
    strncpy(buf_host, llmc::g_last_error, n - 1);
    buf_host[n - 1] = 0;
    return (int)strlen(buf_host);
}
