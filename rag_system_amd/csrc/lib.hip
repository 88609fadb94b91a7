// lib.hip -- process-level state of librag_mi355x.so: device selection, error text, options.
#include "common.h"

#include <cstdlib>
#include <cstring>

namespace rag {

static thread_local std::string g_err;
static std::mutex g_init_mu;
static bool g_init = false;
static int g_dev = -1;
static int g_cus = 256;
static hipStream_t g_stream = nullptr;
static Options g_opt;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

bool initialized() { return g_init; }
int device_id() { return g_dev; }
hipStream_t lib_stream() { return g_stream; }
int num_cus() { return g_cus; }
Options& options() { return g_opt; }

}  // namespace rag

extern "C" {

int rag_abi_version(void) { return 2; }

const char* rag_last_error(void) { return rag::g_err.c_str(); }

int rag_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rag_init(int device_id) {
    std::lock_guard<std::mutex> lk(rag::g_init_mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return rag::fail(RAG_ERR_NO_DEVICE,
                         "no HIP device visible (%s); this engine has no CPU fallback",
                         e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return rag::fail(RAG_ERR_INVALID_ARG, "device_id %d out of range [0,%d)", device_id, n);
    if (rag::g_init) {
        if (rag::g_dev == device_id) return RAG_OK;
        return rag::fail(RAG_ERR_INVALID_ARG,
                         "already initialised on device %d (one process per GPU)", rag::g_dev);
    }
    RAG_HIP(hipSetDevice(device_id));
    // RAG_SYNC_MODE=spin|yield|block (experiment switch): how hipStreamSynchronize waits on this device
    if (const char* m = std::getenv("RAG_SYNC_MODE")) {
        unsigned flags = !std::strcmp(m, "spin") ? hipDeviceScheduleSpin : !std::strcmp(m, "yield") ? hipDeviceScheduleYield
                         : !std::strcmp(m, "block") ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
        (void)hipSetDeviceFlags(flags);
        (void)hipGetLastError();
    }
    hipDeviceProp_t prop;
    RAG_HIP(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rag::fail(RAG_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                         device_id, prop.gcnArchName);
    rag::g_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    RAG_HIP(hipStreamCreateWithFlags(&rag::g_stream, hipStreamNonBlocking));
    {
        const int rc = rag::flat_init_scratch();  // with device_id current
        if (rc) return rc;
    }
    rag::g_dev = device_id;
    rag::g_init = true;
    return RAG_OK;
}

int rag_lib_stream(void** out) {
    RAG_REQUIRE_INIT();
    if (!out) return rag::fail(RAG_ERR_INVALID_ARG, "out is NULL");
    *out = (void*)rag::g_stream;
    return RAG_OK;
}

int rag_set_option(const char* name, int64_t value) {
    if (!name) return rag::fail(RAG_ERR_INVALID_ARG, "option name is NULL");
    rag::Options& o = rag::options();
    if (!std::strcmp(name, "scan_blocks")) o.scan_blocks = value;
    else if (!std::strcmp(name, "scan_nt")) o.scan_nt = value;
    else if (!std::strcmp(name, "scan_rows")) o.scan_rows = value;
    else if (!std::strcmp(name, "mfma_min_nq")) o.mfma_min_nq = value;
    else if (!std::strcmp(name, "mfma_shape")) o.mfma_shape = value;
    else if (!std::strcmp(name, "mfma_bpc")) o.mfma_bpc = value;
    else if (!std::strcmp(name, "mfma_stages")) o.mfma_stages = value;
    else if (!std::strcmp(name, "mfma_filter_k")) o.mfma_filter_k = value;
    else if (!std::strcmp(name, "mfma_filter_min_rows")) o.mfma_filter_min_rows = value;
    else if (!std::strcmp(name, "mfma_filter_step")) o.mfma_filter_step = value;
    else if (!std::strcmp(name, "stream_max_nq")) o.stream_max_nq = value;
    else if (!std::strcmp(name, "stream_rg")) o.stream_rg = value;
    else if (!std::strcmp(name, "stream_bpc")) o.stream_bpc = value;
    else if (!std::strcmp(name, "stream_nb")) o.stream_nb = value;
    else if (!std::strcmp(name, "stream_fuse")) o.stream_fuse = value;
    else if (!std::strcmp(name, "sparse_rank_one")) o.sparse_rank_one = value;
    else if (!std::strcmp(name, "wide_nb")) o.wide_nb = value;
    else if (!std::strcmp(name, "prefilter_kp")) o.prefilter_kp = value;
    else if (!std::strcmp(name, "prefilter_min_rows")) o.prefilter_min_rows = value;
    else if (!std::strcmp(name, "mfma_pipe_sample")) o.mfma_pipe_sample = value;
    else if (!std::strcmp(name, "wide_max_nq")) o.wide_max_nq = value;
    else if (!std::strcmp(name, "wide_min_nq")) o.wide_min_nq = value;
    else if (!std::strcmp(name, "wide_wpb")) o.wide_wpb = value;
    else if (!std::strcmp(name, "sparse_topk_lists")) o.sparse_topk_lists = value;
    else if (!std::strcmp(name, "sparse_fused")) o.sparse_fused = value;
    else return rag::fail(RAG_ERR_INVALID_ARG, "unknown option '%s'", name);
    return RAG_OK;
}

}  // extern "C"
