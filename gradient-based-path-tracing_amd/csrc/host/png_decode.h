// png_decode.h — PNG decoder whose 8-bit output equals stb_image v2.27's for 1 or 3 requested channels (see png_decode.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace gdpt {

// req_comp = 3: interleaved RGB; req_comp = 1: what stbi_load(..., 1) returns (grey, or the integer luma of RGB).
// All colour types, bit depths 1..16, Adam7 interlacing, tRNS. Throws std::runtime_error on malformed input.
void decode_png(const uint8_t *bytes, size_t size, int req_comp, int *width, int *height, std::vector<uint8_t> *out);

} // namespace gdpt
