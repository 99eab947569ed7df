// device side of profiles/aql_probe.cpp: compiled --cuda-device-only into a bare code object
#include <hip/hip_runtime.h>
struct SpinArgs { long long ticks; long long* stamp; };
extern "C" __global__ void probe_spin(SpinArgs a) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) a.stamp[0] = t0;
    while (wall_clock64() - t0 < a.ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && blockIdx.x == 0) a.stamp[1] = wall_clock64();
}
