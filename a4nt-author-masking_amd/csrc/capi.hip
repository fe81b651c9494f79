// C-ABI plumbing: error string, launch check, device queries.  See include/a4nt_hip.h.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

static thread_local char g_err[512] = "";

void a4nt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int a4nt_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        a4nt_set_error("%s: %s", what, hipGetErrorString(e));
        return A4NT_ERR_LAUNCH;
    }
    return A4NT_OK;
}

extern "C" const char* a4nt_last_error(void) { return g_err; }

extern "C" int a4nt_version(void) { return 100; }

// Checks that `device` is a gfx950 part with the CU count the persistent kernels are sized for.
extern "C" int a4nt_init(int device, int* num_cus) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) {
        a4nt_set_error("a4nt_init: cannot query device %d", device);
        return A4NT_ERR_ARG;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        a4nt_set_error("a4nt_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, p.gcnArchName);
        return A4NT_ERR_ARG;
    }
    if (num_cus) *num_cus = p.multiProcessorCount;
    if (p.multiProcessorCount < 256) {
        a4nt_set_error("a4nt_init: device %d exposes %d CUs; the persistent LSTM sweeps need 256 co-resident workgroups", device, p.multiProcessorCount);
        return A4NT_ERR_ARG;
    }
    // every workgroup of a persistent sweep must be resident at once: record what the occupancy calculator says
    // about each instantiation (needs `device` current for the query)
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return a4nt_check_launch("a4nt_init set device");
    const int rc = a4nt_lstm_probe_occupancy(device, p.multiProcessorCount);
    (void)hipSetDevice(prev);
    return rc;
}

// Reads the timeout word a persistent sweep raises when a bounded inter-workgroup wait expired.
// `sync` is the device sync block handed to a4nt_lstm_seq_fwd/bwd, `groups` its group count.
// Synchronises the stream (call it where the host syncs anyway).
extern "C" int a4nt_lstm_check_timeout(const unsigned* sync, int groups, void* stream) {
    unsigned w = 0;
    if (hipMemcpyAsync(&w, sync + 32 * groups, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        if (a4nt_check_launch("lstm_check_timeout") == A4NT_OK) a4nt_set_error("lstm_check_timeout: reading the timeout word failed");
        return A4NT_ERR_LAUNCH;
    }
    if (w != 0) {
        a4nt_set_error("persistent LSTM sweep: inter-workgroup wait timed out (workgroups not co-resident?)");
        return A4NT_ERR_TIMEOUT;
    }
    return A4NT_OK;
}
