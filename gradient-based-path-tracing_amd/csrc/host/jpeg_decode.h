// jpeg_decode.h — baseline + progressive JPEG decoder whose 8-bit output equals stb_image v2.27's (see jpeg_decode.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace gdpt {

// req_comp = 3: interleaved RGB; req_comp = 1: what stbi_load(..., 1) returns (the Y plane for YCbCr files).
// Throws std::runtime_error on malformed or unsupported (12-bit, arithmetic-coded, CMYK) input.
void decode_jpeg(const uint8_t *bytes, size_t size, int req_comp, int *width, int *height, std::vector<uint8_t> *out);

} // namespace gdpt
