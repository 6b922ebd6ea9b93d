// rade_rx2.hip -- translation unit of k_rx_sync2, the receiver kernel with two streams per CU (see rade_rx2.inc).
// A separate code object on purpose: k_rx_sync's register allocation is fragile, and with both kernels in one
// translation unit its spill count went from 99 to 123 and its launch from 4.20 to 4.53 ms.  The device helpers both kernels
// share (FFT correlator pieces, refine(), decoder-stage descriptors, reductions) come from rade_kernels.hip with its kernels and
// launch shims compiled out.
#define RADE_RX2_TU 1
#include "rade_kernels.hip"
#include "rade_rx2.inc"

extern "C" int rd_launch_rx_sync2(const rd_sync_args *a, rd_stream_t s)
{
    if (a->B <= 0) return 0;
    static int lds2[64];                          // per device; RADE_RX2_SOLO=1 (developer switch) asks for more than half the LDS: one workgroup per CU
    int dev_ = 0; (void)hipGetDevice(&dev_);
    int &l = lds2[dev_ & 63];
    if (!l) { l = getenv("RADE_RX2_SOLO") ? 100 * 1024 : (int)sizeof(RxShared2); (void)hipFuncSetAttribute((const void *)k_rx_sync2, hipFuncAttributeMaxDynamicSharedMemorySize, l); }
    hipLaunchKernelGGL(k_rx_sync2, dim3(a->B), dim3(NT2), l, (hipStream_t)s, *a);
    return (int)hipGetLastError();
}
