// Test-infrastructure stand-in for lodepng.h (not on disk; a third-party dependency of the
// reference's model compiler).  PNG pixels are NOT decoded: a file with a valid PNG signature and
// IHDR chunk yields a blank (all-zero) image of the header's width x height, anything else fails.
// Textures do not take part in mj_step; this only lets models that reference PNG textures
// (model/cube/cube_3x3x3.xml) compile in the oracle build.
#ifndef ORACLE_STUB_LODEPNG_H_
#define ORACLE_STUB_LODEPNG_H_
#include <cstddef>
#include <cstdlib>
#include <cstring>
enum LodePNGColorType { LCT_GREY = 0, LCT_RGB = 2, LCT_PALETTE = 3, LCT_GREY_ALPHA = 4, LCT_RGBA = 6 };
struct LodePNGColorMode { LodePNGColorType colortype; unsigned bitdepth; };
struct LodePNGInfo { unsigned srgb_defined; };
namespace lodepng {
struct State { LodePNGColorMode info_raw{LCT_RGBA, 8}; LodePNGInfo info_png{0}; };
}
inline size_t lodepng_get_raw_size(unsigned w, unsigned h, const LodePNGColorMode* mode) {
  size_t channels = 4;
  switch (mode->colortype) {
    case LCT_GREY: case LCT_PALETTE: channels = 1; break;
    case LCT_GREY_ALPHA: channels = 2; break;
    case LCT_RGB: channels = 3; break;
    default: channels = 4; break;
  }
  return (size_t)w * h * channels * (mode->bitdepth / 8 ? mode->bitdepth / 8 : 1);
}
inline unsigned lodepng_decode(unsigned char** out, unsigned* w, unsigned* h, lodepng::State* state,
                               const unsigned char* in, size_t insize) {
  static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  *out = nullptr; *w = 0; *h = 0;
  if (insize < 24 || memcmp(in, sig, 8) != 0 || memcmp(in + 12, "IHDR", 4) != 0) return 1;
  *w = ((unsigned)in[16] << 24) | ((unsigned)in[17] << 16) | ((unsigned)in[18] << 8) | in[19];
  *h = ((unsigned)in[20] << 24) | ((unsigned)in[21] << 16) | ((unsigned)in[22] << 8) | in[23];
  if (*w == 0 || *h == 0 || *w > 16384 || *h > 16384) return 1;
  *out = (unsigned char*)calloc(lodepng_get_raw_size(*w, *h, &state->info_raw), 1);
  return *out ? 0 : 1;
}
inline const char* lodepng_error_text(unsigned) { return "PNG decoding unavailable in the oracle build"; }
#endif
