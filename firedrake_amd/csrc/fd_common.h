// fd_common.h -- internal helpers shared by the libfdhip.so translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <string>
#include "../../include/fdhip.h"

namespace fd {
void set_error(const std::string &msg);
// A NULL stream argument means "the library's default stream": the HIP null stream, or the capturing
// stream while a hipGraph of an assembly step is being recorded (fd_graph_begin).
hipStream_t default_stream();
inline hipStream_t st(fd_stream_t s) { return s ? reinterpret_cast<hipStream_t>(s) : default_stream(); }
// hipFree for the release entry points (fd_free, fd_*_free): carried out at fd_graph_end when it arrives while a step is being
// captured -- hipFree synchronises the device, which invalidates the capture, and a finaliser can run at any moment
hipError_t release(void *p);
}  // namespace fd

#define FD_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess) {                                                        \
            fd::set_error(std::string(#call) + " failed: " + hipGetErrorString(_e) +   \
                          " (" __FILE__ ":" + std::to_string(__LINE__) + ")");         \
            return (int)_e ? (int)_e : -1;                                             \
        }                                                                              \
    } while (0)

#define FD_FAIL(msg)                \
    do {                            \
        fd::set_error(msg);         \
        return -1;                  \
    } while (0)

#define FD_CHECK_LAUNCH() FD_HIP(hipGetLastError())

// position of column c in CSR row r (sorted columns), or -1: row starts are fd_nnz_t, the place inside a row fits an int
__device__ inline fd_nnz_t fd_csr_find(const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int r, int c) {
    fd_nnz_t lo = rowptr[r], hi = rowptr[r + 1] - 1;
    while (lo <= hi) {
        const fd_nnz_t mid = lo + ((hi - lo) >> 1);
        const int v = colidx[mid];
        if (v == c) return mid;
        if (v < c) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}
