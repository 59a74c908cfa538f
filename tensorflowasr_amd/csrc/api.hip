// Status strings / ABI version of libtfasr_hip.so (include/tfasr_hip.h).
#include "common.h"

extern "C" const char* tfasr_status_string(int status) {
  switch (status) {
    case TFASR_STATUS_SUCCESS: return "success";
    case TFASR_STATUS_INVALID_VALUE: return "invalid value";
    case TFASR_STATUS_EXECUTION_FAILED: return "kernel launch / execution failed";
    case TFASR_STATUS_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}

extern "C" int tfasr_abi_version(void) { return TFASR_ABI_VERSION; }

std::atomic<size_t> g_tfasr_launch_count{0};
extern "C" size_t tfasr_launch_count(void) { return g_tfasr_launch_count.load(std::memory_order_relaxed); }
