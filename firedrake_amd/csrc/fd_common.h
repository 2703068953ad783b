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
