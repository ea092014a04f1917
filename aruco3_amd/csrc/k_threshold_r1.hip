// k_threshold_r1.hip -- instantiations of the register-resident threshold kernel for radii 1..3 (see k_threshold_k1.h)
#include "k_threshold_k1.h"

namespace a3 {

hipError_t launch_k1_r1(uint32_t radius, hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H, uint32_t n,
                        uint8_t* grey, uint64_t* bits) {
    switch (radius) {
        case 1: return launch_k1<1>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 2: return launch_k1<2>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 3: return launch_k1<3>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace a3
