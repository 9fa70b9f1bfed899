// Instrumented build only (`make stamp`, -DYOLO_STAMP; never loaded by the package): per-block shader-clock stamps of the phases of
// conv_pipe_kernel (the 3x3 translation unit) and, slots 8-15, of the conv epilogue as wave 0 of the block walks it
// (tools/stamp_probe.py).  A persistent block keeps the stamps of its LAST tile.  Everywhere else STAMP() is nothing.
#pragma once
#if defined(YOLO_STAMP) && defined(YOLO_PIPE_PART) && YOLO_PIPE_PART == 0
#define YOLO_STAMP_ON 1
#define YOLO_STAMP_SLOTS 16
__device__ long long yolo_stamps[8192 * YOLO_STAMP_SLOTS];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) yolo_stamps[blockIdx.x * YOLO_STAMP_SLOTS + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#define STAMP_ID() do { if (threadIdx.x == 0 && blockIdx.x < 8192) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); yolo_stamps[blockIdx.x * YOLO_STAMP_SLOTS + 7] = (long long)id | ((long long)(xcc & 0xf) << 32); yolo_stamps[blockIdx.x * YOLO_STAMP_SLOTS + 6] = wall_clock64(); } } while (0)
extern "C" int yolo_debug_read_stamps(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(yolo_stamps), sizeof(long long) * n, 0, hipMemcpyDeviceToHost);
}
#else
#define STAMP(i)
#define STAMP_ID()
#endif
