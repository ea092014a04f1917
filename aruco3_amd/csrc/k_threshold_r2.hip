// k_threshold_r2.hip -- instantiations of the register-resident threshold kernel for radii 4..6 (see k_threshold_k1.h)
#include "k_threshold_k1.h"

namespace a3 {

hipError_t launch_k1_r2(uint32_t radius, hipStream_t st, const uint8_t* pixels, int fmt, size_t row_stride, size_t frame_stride, int W, int H, uint32_t n,
                        uint8_t* grey, uint64_t* bits) {
    switch (radius) {
        case 4: return launch_k1<4>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 5: return launch_k1<5>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        case 6: return launch_k1<6>(st, pixels, fmt, row_stride, frame_stride, W, H, n, grey, bits);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace a3
