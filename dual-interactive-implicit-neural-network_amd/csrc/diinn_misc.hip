// diinn_misc.hip -- error state of the launch functions and the device sine / axis-table test hooks
// (part of libdiinn_hip.so; shared definitions in diinn_device.h, layout in diinn_layout.h)
#include "diinn_device.h"

// ---------------------------------------------------------------------------------
// sine kernel (tests): the device sine of each mode, elementwise
// ---------------------------------------------------------------------------------
template <int MODE>
__global__ void sin_kernel(const float* x, float* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = dsin<MODE>(x[i]);
}

// ---------------------------------------------------------------------------------
// tables kernel (tests): the device evaluation of axis_eval
// ---------------------------------------------------------------------------------
__global__ void axis_tables_kernel(Axis a, int n_out, int32_t* idx, float* rel) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_out) return;
    int id;
    float r;
    axis_eval(a, j, id, r);
    if (idx) idx[j] = id;
    if (rel) rel[j] = r;
}

// ---------------------------------------------------------------------------------
// clock probe (diagnosis; tools/clock_trace.py): ONE wave that runs beside whatever else the chip is doing and samples the
// SHADER clock the part holds: per sample, the s_memtime ticks (shader cycles) that passed while s_memrealtime (the constant
// 100 MHz counter) advanced by `ticks` -- MI355X_MICROARCH.md, DVFS give-back item 6: clock = d(memtime) / d(memrealtime) x
// 100 MHz.  The wave sleeps between looks; it writes only its own sample buffer.
// ---------------------------------------------------------------------------------
__global__ void clock_probe_kernel(unsigned long long* out, int n, unsigned ticks) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        unsigned long long r0, r1, c0, c1;
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(c0)::"memory");
        do {
            __builtin_amdgcn_s_sleep(64);
            asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1), "=s"(c1)::"memory");
        } while (r1 - r0 < ticks);
        out[3 * i] = c1 - c0;
        out[3 * i + 1] = r1 - r0;
        out[3 * i + 2] = r0;
    }
}

thread_local int g_last_hip_error = 0;

// compute units of the current device, asked once per device (cost models and persistent grids: an MI355X has 256, a
// partitioned one fewer); 256 if the runtime cannot say
int device_cus() {
    const long long forced = knob(diinn_knobs().debug_ncu);      // tests: a fixed count whatever the box (DIINN_DEBUG_NCU)
    if (forced > 0) return (int)(forced < 65536 ? forced : 65536);
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
#ifdef DIINN_STAMPS
unsigned long long* g_stamps = nullptr;
extern "C" int diinn_debug_set_stamp_buffer(void* dev_ptr) { g_stamps = (unsigned long long*)dev_ptr; return 0; }
#endif

extern "C" {

int diinn_last_hip_error(void) { return g_last_hip_error; }

int diinn_debug_clock_probe(void* stream, unsigned long long* samples_dev, int n, unsigned realtime_ticks) {
    if (!samples_dev || n <= 0 || realtime_ticks == 0) return DIINN_ERR_INVALID_ARG;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, samples_dev, n, realtime_ticks);
    return hip_status(hipGetLastError());
}

int diinn_make_axis_tables_device(void* stream, int n_in, int n_out, int small_output,
                                  int32_t* idx_dev, float* rel_dev) {
    if (n_in <= 0 || n_out <= 0) return DIINN_ERR_INVALID_ARG;
    const Axis a = make_axis(n_in, n_out, small_output ? 1 : 0);
    hipLaunchKernelGGL(axis_tables_kernel, dim3((n_out + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, a, n_out, idx_dev, rel_dev);
    return hip_status(hipGetLastError());
}

int diinn_eval_sin_device(void* stream, int sin_mode, const float* x_dev, float* y_dev, int n) {
    if (!x_dev || !y_dev || n <= 0) return DIINN_ERR_INVALID_ARG;
    const dim3 grid((n + 255) / 256), blk(256);
    switch (sin_mode) {
        case DIINN_SIN_ACCURATE: hipLaunchKernelGGL(sin_kernel<DIINN_SIN_ACCURATE>, grid, blk, 0, (hipStream_t)stream, x_dev, y_dev, n); break;
        case DIINN_SIN_HW: hipLaunchKernelGGL(sin_kernel<DIINN_SIN_HW>, grid, blk, 0, (hipStream_t)stream, x_dev, y_dev, n); break;
        case DIINN_SIN_HW_REDUCED: hipLaunchKernelGGL(sin_kernel<DIINN_SIN_HW_REDUCED>, grid, blk, 0, (hipStream_t)stream, x_dev, y_dev, n); break;
        default: return DIINN_ERR_UNSUPPORTED;
    }
    return hip_status(hipGetLastError());
}

}  // extern "C"
