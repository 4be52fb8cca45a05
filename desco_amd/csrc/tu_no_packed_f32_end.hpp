// closes tu_no_packed_f32_begin.hpp (last line of the translation unit)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif
