// Wide-tap blur, both passes in one launch over 32-row panels (blur_panel.hip); dispatched from bg_blur_nhwc_f32 (blur.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace bg {
bool blur_panel_ok(int B, int H, int W, int C, int n_taps);
size_t blur_panel_lds_bytes(int W, int n_taps);
double blur_panel_exec_flops(int B, int H, int W, int n_taps);
int blur_panel_launch(const float* x, float* y, int B, int H, int W, const float* taps_d, int n_taps, hipStream_t s);
}  // namespace bg
