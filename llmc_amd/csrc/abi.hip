// abi.hip — version / error entry points of libllmc_hip.so.
#include <mutex>
#include <set>
#include <utility>
#include "common.h"

namespace llmc {
static thread_local char g_last_error[512] = "";
static thread_local int g_helper_streams = 1;
bool helper_streams_enabled() { return g_helper_streams != 0; }
static thread_local int g_cu_reserve = 0;
int cu_reserve() { return g_cu_reserve; }
static thread_local int g_opt[OPT_COUNT] = {};
int opt(int id) { return (id >= 0 && id < OPT_COUNT) ? g_opt[id] : 0; }
static const char* const g_opt_names[OPT_COUNT] = {
    "k3_fp32", "k3_no_gemm6", "k3_no_planes", "k3_split_far", "k4_split_far", "k4_err_rowmajor", "gptq_generic",
    "gemm3_nospec", "gemm3s_min_tiles", "no_shortk", "linear_nosplit", "fp8_exact_div", "side_cu_mask", "k1_batch_off", "k1_fp32_diag", "fp8_no_packed16", "gemm3s_no_dma", "sgemm_no_wide", "gemm3_no_wide"};

void set_last_error(const char* where, hipError_t e) {
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}
void set_last_error_msg(const char* msg) { snprintf(g_last_error, sizeof(g_last_error), "%s", msg); }

// ---- per-device facts and per-(device, kernel) function attributes ---------------------------------------
// The only process-wide mutable state of the library besides the side-stream pool; both are guarded by a mutex so
// that entry points may be called from several host threads (re-entrant).
static std::mutex g_dev_mu;
static int g_cu_count[LLMC_MAX_DEVICES] = {};
static std::set<std::pair<int, const void*>> g_lds_attr_done;

int device_cu_count() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= LLMC_MAX_DEVICES) return 256;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (g_cu_count[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cu_count[dev] = n;
    }
    return g_cu_count[dev];
}

int ensure_dynamic_lds(const void* fn, int bytes) {
    int dev = 0;
    LLMC_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_dev_mu);
    auto key = std::make_pair(dev, fn);
    if (g_lds_attr_done.count(key)) return LLMC_OK;
    LLMC_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_lds_attr_done.insert(key);
    return LLMC_OK;
}
}  // namespace llmc

extern "C" int llmc_hip_set_helper_streams(int enable) {
    int prev = llmc::g_helper_streams;
    llmc::g_helper_streams = enable ? 1 : 0;
    return prev;
}

extern "C" int llmc_hip_set_option(const char* key, int value) {
    if (!key || value < 0) return LLMC_EINVAL;
    for (int i = 0; i < llmc::OPT_COUNT; ++i)
        if (strcmp(key, llmc::g_opt_names[i]) == 0) {
            const int prev = llmc::g_opt[i];
            llmc::g_opt[i] = value;
            return prev;
        }
    llmc::set_last_error_msg("llmc_hip_set_option: unknown key");
    return LLMC_EINVAL;
}

extern "C" int llmc_hip_get_option(const char* key) {
    if (!key) return LLMC_EINVAL;
    for (int i = 0; i < llmc::OPT_COUNT; ++i)
        if (strcmp(key, llmc::g_opt_names[i]) == 0) return llmc::g_opt[i];
    return LLMC_EINVAL;
}

extern "C" int llmc_hip_option_name(int index, char* buf_host, size_t n) {
    if (index < 0 || index >= llmc::OPT_COUNT || !buf_host || n == 0) return LLMC_EINVAL;
    strncpy(buf_host, llmc::g_opt_names[index], n - 1);
    buf_host[n - 1] = 0;
    return (int)strlen(buf_host);
}

extern "C" int llmc_hip_set_cu_reserve(int n_cus) {
    int prev = llmc::g_cu_reserve;
    llmc::g_cu_reserve = n_cus < 0 ? 0 : n_cus;
    return prev;
}

extern "C" int llmc_hip_abi_version(void) { return LLMC_HIP_ABI_VERSION; }

// hash of the sources and flags this library was built from (llmc_amd/build.py:source_digest), checked by the loader
#ifndef LLMC_BUILD_ID
#define LLMC_BUILD_ID "unknown"
#endif
extern "C" const char* llmc_hip_build_id(void) { return LLMC_BUILD_ID; }

extern "C" int llmc_hip_last_error(char* buf_host, size_t n) {
    if (!buf_host || n == 0) return (int)strlen(llmc::g_last_error);
    strncpy(buf_host, llmc::g_last_error, n - 1);
    buf_host[n - 1] = 0;
    return (int)strlen(buf_host);
}
