// Diagnostic kernel (bench.py only): read `bytes` once with the same access pattern as the GEMV weight stream
// (16 B/lane, non-temporal, 4 loads in flight per lane) and do nothing else.  Its duration is the floor any kernel
// that must read that many bytes from HBM can reach on this chip: the yardstick next to roofline.frac.
#include "gemv_kernel.hpp"

namespace eetq {

namespace {
__global__ __launch_bounds__(1024) void stream_read_kernel(const u32x4* __restrict__ p, unsigned* __restrict__ sink)
{
    const u32x4* q = p + (size_t)blockIdx.x * 4096 + threadIdx.x;
    u32x4        v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gemv::load_w<true>(q + i * 1024);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x9e3779b9u) sink[0] = acc;  // practically never true: keeps the loads alive
}
// touches no memory: what a dispatch with the GEMV's launch geometry costs before it moves a byte
__global__ __launch_bounds__(1024) void empty_kernel(unsigned* sink, int never)
{
    if (never == 12345) sink[0] = threadIdx.x;
}
// one wave per workgroup; lane 0 records where it ran and both device clocks, read back to back:
//   out[4b + 0] = XCC_ID (the XCD the workgroup landed on), [1] = HW_ID, [2] = s_memtime (one tick = one SHADER cycle,
//   MI355X_MICROARCH.md), [3] = s_memrealtime (constant 100 MHz).  Two such launches around a chain of kernels give, per XCD,
//   d(memtime) / d(memrealtime) x 100 MHz = the average shader clock the chip held over the chain.
__global__ __launch_bounds__(64) void clock_stamp_kernel(unsigned long long* __restrict__ out)
{
    if (threadIdx.x != 0) return;
    unsigned long long tm, tr;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm), "=s"(tr)::"memory");
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);    // hwreg(HW_REG_XCC_ID, 0, 4)
    const unsigned hw  = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);    // hwreg(HW_REG_HW_ID)
    unsigned long long* o = out + (size_t)blockIdx.x * 4;
    o[0] = xcc;
    o[1] = hw;
    o[2] = tm;
    o[3] = tr;
}
}  // namespace

int launch_clock_stamp(unsigned long long* out, int grid, hipStream_t stream)
{
    EETQ_REQUIRE(out && grid > 0 && grid <= 65536, "clock stamp: invalid arguments");
    launch_kernel(clock_stamp_kernel, dim3((unsigned)grid), dim3(64), 0, stream, out);
    return check_hip(hipGetLastError(), "clock_stamp_kernel launch");
}

int launch_empty(unsigned* sink, int grid, int block, hipStream_t stream)
{
    EETQ_REQUIRE(sink && grid > 0 && block > 0 && block <= 1024, "empty kernel: invalid launch geometry");
    launch_kernel(empty_kernel, dim3((unsigned)grid), dim3((unsigned)block), 0, stream, sink, 0);
    return check_hip(hipGetLastError(), "empty_kernel launch");
}

int launch_stream_read(const void* p, size_t bytes, unsigned* sink, hipStream_t stream)
{
    EETQ_REQUIRE(p && sink && bytes >= 65536 && bytes % 65536 == 0, "stream_read: bytes must be a multiple of 64 KiB");
    launch_kernel(stream_read_kernel, dim3((unsigned)(bytes / 65536)), dim3(1024), 0, stream,
                  static_cast<const u32x4*>(p), sink);
    return check_hip(hipGetLastError(), "stream_read_kernel launch");
}

}  // namespace eetq
