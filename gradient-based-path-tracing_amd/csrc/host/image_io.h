// image_io.h — output writers and texture input for the GradPath boundary.
// Mirrors imwrite()/imread1()/imread3() of the reference (src/image.cpp:26-173).
#pragma once
#include <string>
#include <vector>

namespace gdpt {

// ".pfm": "PF\nW H\n-1\n" + fp32 RGB rows exactly as stored (top row first; src/image.cpp:141-149).
// ".exr": RGB half-float scanline file (the reference calls tinyexr SaveEXR(..., 3, 1 /*fp16*/), :155-171).
// Any other suffix writes nothing, like the reference.
void write_image(const std::string &filename, int width, int height, const double *rgb);

// Reads a PFM written by write_image (tests, CLI round trips).
void read_pfm(const std::string &filename, int *width, int *height, std::vector<double> *rgb);

// Texture input for <texture type="bitmap">. The hot path needs no codec: `path` is used if it is a
// .pfm, otherwise the pre-decoded companion `path + ".gdtex"` is read
// (GDTEX1: "GDTEX1\n", int32 w,h,c, w*h*c fp32 texels holding exactly what stbi_loadf/LoadEXR would return;
// GDTEX2: "GDTEX2\n", int32 w,h,c, uint32 zlen, zlib(uint8 texels), widened with stb_image's 8-bit -> float rule;
// written by scenes/tools/predecode_textures.py).
// `channels` = 1 (imread1) or 3 (imread3); a 3-channel source is reduced the way imread1 does for EXR
// ((r+g+b)/3, src/image.cpp:63-65) only for .pfm/.exr-derived data; 8-bit sources are reduced by the
// generator script with stb's integer luma so that texels are already 1-channel.
void load_texture_file(const std::string &path, int channels, int *width, int *height, std::vector<double> *texels);

} // namespace gdpt
